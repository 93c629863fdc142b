// Gather lab 2 (round 2): isolates what limits the table-driven walk (walk_fat_kernel, 2.6e10 steps/s on C3).
// Lab 1 showed no cliff with table size (16-B gathers 4.9e10/s at 56 GB, 5.6e10/s from the Infinity Cache) but
// the 32-B slot read as two dwordx4 loads at 3.8e10/s.  Here: a cheap dependent chain (LCG + mulhi range
// reduction, no 64-bit modulo), and per 32-B slot
//   V_TWO   two dwordx4 loads per lane (what walk_fat_kernel does),
//   V_PAIR  lane pairs load the two halves of ONE slot with ONE load instruction each (32 contiguous bytes per
//           pair -> one request), halves exchanged through DPP,
// each with / without the walk's output stream (4 B per step per lane, written in 16-B or 64-B pieces).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum { V_ONE16 = 0, V_TWO = 1, V_PAIR = 2 };

__device__ __forceinline__ uint4 shfl_xor1(uint4 v) {
    uint4 r;
    r.x = __shfl_xor((int)v.x, 1, 64); r.y = __shfl_xor((int)v.y, 1, 64);
    r.z = __shfl_xor((int)v.z, 1, 64); r.w = __shfl_xor((int)v.w, 1, 64);
    return r;
}

template <int VAR, int WR>   // WR: 0 no output, 4 = 16-B pieces (4 steps), 16 = 64-B pieces (16 steps)
__global__ void __launch_bounds__(256) chain(const uint8_t* __restrict__ tab, uint64_t n_slots, int steps, int32_t* __restrict__ walks) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t state = (g + 1) * 0x9e3779b97f4a7c15ULL;
    uint64_t idx = __umul64hi(state, n_slots);
    uint32_t acc = 0;
    int32_t* out = walks + g * (uint64_t)steps;
    int32_t buf[WR ? WR : 1];
    for (int s = 0; s < steps; ++s) {
        uint32_t v;
        if (VAR == V_ONE16) {
            const uint4 x = *reinterpret_cast<const uint4*>(tab + idx * 16);
            v = x.x ^ x.y ^ x.z ^ x.w;
        } else if (VAR == V_TWO) {
            const uint4* p = reinterpret_cast<const uint4*>(tab + idx * 32);
            const uint4 x = p[0], y = p[1];
            v = x.x ^ x.y ^ x.z ^ x.w ^ y.x ^ y.y ^ y.z ^ y.w;
        } else {
            const int odd = threadIdx.x & 1;
            const uint64_t other = ((uint64_t)(uint32_t)__shfl_xor((int)(idx >> 32), 1, 64) << 32) | (uint32_t)__shfl_xor((int)idx, 1, 64);
            const uint64_t even_idx = odd ? other : idx, odd_idx = odd ? idx : other;
            const uint4 a = *reinterpret_cast<const uint4*>(tab + even_idx * 32 + odd * 16);   // the pair reads 32 contiguous bytes
            const uint4 b = *reinterpret_cast<const uint4*>(tab + odd_idx * 32 + odd * 16);
            const uint4 give = odd ? a : b, got = shfl_xor1(give);
            const uint4 mine = odd ? b : a;
            v = mine.x ^ mine.y ^ mine.z ^ mine.w ^ got.x ^ got.y ^ got.z ^ got.w;
        }
        acc += v;
        state = state * 6364136223846793005ULL + 1442695040888963407ULL + v;
        idx = __umul64hi(state, n_slots);
        if (WR) {
            buf[s % WR] = (int32_t)(idx & 0xfffff);
            if ((s % WR) == WR - 1) {
#pragma unroll
                for (int k = 0; k < WR; k += 4)
                    *reinterpret_cast<int4*>(out + s - (WR - 1) + k) = make_int4(buf[k], buf[k + 1], buf[k + 2], buf[k + 3]);
            }
        }
    }
    if (acc == 0x12345678u) walks[0] = (int32_t)acc;
}

template <int VAR, int WR>
static double run(const uint8_t* tab, uint64_t bytes, int blocks, int steps, int32_t* walks) {
    const uint64_t n_slots = bytes / (VAR == V_ONE16 ? 16 : 32);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((chain<VAR, WR>), dim3(blocks), dim3(256), 0, 0, tab, n_slots, steps, walks);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    return (double)blocks * 256.0 * steps / (best * 1e-3);
}

int main(int argc, char** argv) {
    const double max_gb = argc > 1 ? atof(argv[1]) : 56.0;
    const size_t cap = (size_t)(max_gb * (1ull << 30));
    const int blocks = 256 * 8 * 16, steps = 80;   // 8.4M lanes x 80 steps, like one C3 round
    uint8_t* tab; int32_t* walks;
    CK(hipMalloc(&tab, cap)); CK(hipMalloc(&walks, (size_t)blocks * 256 * steps * 4 + 64));
    CK(hipMemset(tab, 0, cap)); CK(hipDeviceSynchronize());
    for (double gb : {1.0, 16.0, 56.0}) {
        if (gb > max_gb) break;
        const uint64_t bytes = (uint64_t)(gb * (1ull << 30));
        printf("{\"gb\":%.0f,\"one16\":%.4g,\"two32\":%.4g,\"pair32\":%.4g,\"two32_w16B\":%.4g,\"pair32_w16B\":%.4g,\"two32_w64B\":%.4g,\"pair32_w64B\":%.4g,\"one16_w64B\":%.4g}\n",
               gb, run<V_ONE16, 0>(tab, bytes, blocks, steps, walks), run<V_TWO, 0>(tab, bytes, blocks, steps, walks),
               run<V_PAIR, 0>(tab, bytes, blocks, steps, walks), run<V_TWO, 4>(tab, bytes, blocks, steps, walks),
               run<V_PAIR, 4>(tab, bytes, blocks, steps, walks), run<V_TWO, 16>(tab, bytes, blocks, steps, walks),
               run<V_PAIR, 16>(tab, bytes, blocks, steps, walks), run<V_ONE16, 16>(tab, bytes, blocks, steps, walks));
        fflush(stdout);
    }
    return 0;
}
