// Shared host-side helpers for the C-ABI implementation (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "n2v_hip.h"

namespace n2v {

inline char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(N2V_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return N2V_OK;
}

inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

// ---- device helpers shared by the walk and table kernels -----------------------------------------

// Philox4x32-10 (Salmon et al., SC'11), key = seed, counter = (walk lo, walk hi, step, 0): the two uniforms of
// step `step` of the walk with GLOBAL index `walk` in throughput mode.  53-bit doubles built exactly like
// MT19937's genrand_res53 (numpy random_sample), so they play the role of the two np.random.rand() calls of
// alias_draw (src/node2vec.py:277-278).
__device__ __forceinline__ void philox_uniforms(uint64_t seed, uint64_t walk, uint32_t step, double& u1, double& u2) {
    uint32_t c0 = (uint32_t)walk, c1 = (uint32_t)(walk >> 32), c2 = step, c3 = 0u;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    u1 = ((double)(c0 >> 5) * 67108864.0 + (double)(c1 >> 6)) / 9007199254740992.0;
    u2 = ((double)(c2 >> 5) * 67108864.0 + (double)(c3 >> 6)) / 9007199254740992.0;
}

// G.has_edge(u, v) on the sorted CSR (src/node2vec.py:145): binary search of v in row u.
__device__ __forceinline__ bool row_contains(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                             int32_t u, int32_t v) {
    int64_t lo = row_ptr[u];
    const int64_t end = row_ptr[u + 1];
    int64_t hi = end;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (col[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo < end && col[lo] == v;
}

}  // namespace n2v
