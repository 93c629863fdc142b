// Gather-rate lab for the walk design (round 2): how many dependent random gathers per second does one MI355X
// sustain as a function of table footprint, bytes per gather, occupancy and address locality?
// Stand-alone executable (hipcc --offload-arch=gfx950 gather_lab.hip -o gather_lab); prints one JSON line per case.
// Every lane runs a DEPENDENT chain like a walk: the next index is a hash of (lane state, loaded value).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL; x ^= x >> 27; x *= 0x94d049bb133111ebULL; x ^= x >> 31;
    return x;
}

// BYTES per gather: 8, 16, 32 (32 = two dwordx4 of one 32-B aligned slot, as walk_fat_kernel does).
// MODE 0: uniformly random over the whole table.  MODE 1: "sorted" - lane g's accesses stay inside a window of
// `window` elements around (g / n_threads) * n_elems (what a per-step sort by table address would give).
// MODE 2: oscillation - with probability 1/2 the lane returns to the element it read two steps ago (+- 4 elements).
template <int BYTES, int MODE>
__global__ void __launch_bounds__(256) gather_chain(const uint8_t* __restrict__ tab, uint64_t n_elems, int steps,
                                                    uint64_t window, uint32_t* out) {
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t n_threads = (uint64_t)gridDim.x * 256;
    uint64_t state = mix64(g + 0x9e3779b97f4a7c15ULL);
    uint64_t idx = state % n_elems, prev1 = idx, prev2 = idx;
    uint32_t acc = 0;
    const uint64_t centre = (uint64_t)((double)g / (double)n_threads * (double)n_elems);
    for (int s = 0; s < steps; ++s) {
        uint32_t v;
        if (BYTES == 8) {
            const uint2 x = *reinterpret_cast<const uint2*>(tab + idx * 8);
            v = x.x ^ x.y;
        } else if (BYTES == 16) {
            const uint4 x = *reinterpret_cast<const uint4*>(tab + idx * 16);
            v = x.x ^ x.y ^ x.z ^ x.w;
        } else {
            const uint4* p = reinterpret_cast<const uint4*>(tab + idx * 32);
            const uint4 x = p[0], y = p[1];
            v = x.x ^ x.y ^ x.z ^ x.w ^ y.x ^ y.y ^ y.z ^ y.w;
        }
        acc += v;
        state = mix64(state + v + 1);   // table is zero-filled; the dependence is real for the compiler and the hardware
        prev2 = prev1; prev1 = idx;
        if (MODE == 0) idx = state % n_elems;
        else if (MODE == 1) { idx = centre + (state % window); if (idx >= n_elems) idx -= n_elems; }
        else { idx = (state & (1ull << 40)) ? (prev2 + ((state >> 41) & 7)) % n_elems : state % n_elems; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int BYTES, int MODE>
static double run(const uint8_t* tab, uint64_t bytes, int blocks, int steps, uint64_t window, uint32_t* out, int lds_pad) {
    const uint64_t n_elems = bytes / BYTES;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((gather_chain<BYTES, MODE>), dim3(blocks), dim3(256), lds_pad, 0, tab, n_elems, steps, window, out);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep > 0 && ms < best) best = ms;
    }
    return (double)blocks * 256.0 * steps / (best * 1e-3);
}

int main(int argc, char** argv) {
    const double max_gb = argc > 1 ? atof(argv[1]) : 56.0;
    size_t cap = (size_t)(max_gb * (1ull << 30));
    uint8_t* tab; uint32_t* out;
    CK(hipMalloc(&tab, cap)); CK(hipMalloc(&out, 64));
    CK(hipMemset(tab, 0, cap)); CK(hipDeviceSynchronize());
    const double sizes_gb[] = {0.125, 1, 4, 8, 16, 28, 56};
    const int steps = 80;
    // occupancy by dynamic LDS: 160 KiB per CU; 256-thread workgroups (4 waves): pad 0 -> 8 wg/CU (32 waves),
    // 40 KiB -> 4 wg (16 waves), 80 KiB -> 2 wg (8 waves)
    const int pads[] = {0, 40 * 1024, 80 * 1024 - 512};
    const int waves[] = {32, 16, 8};
    for (double gb : sizes_gb) {
        if (gb > max_gb) break;
        const uint64_t bytes = (uint64_t)(gb * (1ull << 30));
        for (int oi = 0; oi < 3; ++oi) {
            const int blocks = 256 * 8 * 16;   // 8.4M lanes: 16 waves' worth per slot at full occupancy
            if (oi > 0 && !(gb == 16 || gb == 56)) continue;
            const double r8 = run<8, 0>(tab, bytes, blocks, steps, 0, out, pads[oi]);
            const double r16 = run<16, 0>(tab, bytes, blocks, steps, 0, out, pads[oi]);
            const double r32 = run<32, 0>(tab, bytes, blocks, steps, 0, out, pads[oi]);
            printf("{\"case\":\"random\",\"gb\":%.3f,\"waves_per_cu\":%d,\"g8\":%.4g,\"g16\":%.4g,\"g32\":%.4g}\n", gb, waves[oi], r8, r16, r32);
            fflush(stdout);
        }
        if (gb >= 16) {
            const int blocks = 256 * 8 * 16;
            for (uint64_t win_mb : {2ull, 64ull, 1024ull}) {
                const double r32 = run<32, 1>(tab, bytes, blocks, steps, win_mb * (1 << 20) / 32, out, 0);
                const double r16 = run<16, 1>(tab, bytes, blocks, steps, win_mb * (1 << 20) / 16, out, 0);
                printf("{\"case\":\"sorted\",\"gb\":%.3f,\"window_mb\":%llu,\"g16\":%.4g,\"g32\":%.4g}\n", gb, (unsigned long long)win_mb, r16, r32);
                fflush(stdout);
            }
            const double o8 = run<8, 2>(tab, bytes, blocks, steps, 0, out, 0);
            const double o32 = run<32, 2>(tab, bytes, blocks, steps, 0, out, 0);
            const double o8b = run<8, 2>(tab, bytes, blocks, steps, 0, out, pads[2]);
            const double o32b = run<32, 2>(tab, bytes, blocks, steps, 0, out, pads[2]);
            printf("{\"case\":\"oscillate\",\"gb\":%.3f,\"g8_w32\":%.4g,\"g32_w32\":%.4g,\"g8_w8\":%.4g,\"g32_w8\":%.4g}\n", gb, o8, o32, o8b, o32b);
            fflush(stdout);
        }
    }
    return 0;
}
