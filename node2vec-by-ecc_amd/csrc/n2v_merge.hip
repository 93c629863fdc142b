// Replica merges of the multi-GPU skip-gram trainer — gfx950 (MI355X).  C-ABI and the scheme: include/n2v_hip.h
// ("replica merges").  The reference has no counterpart (gensim's worker threads share ONE table,
// src/main.py:87); across GPUs each rank trains a replica and the replicas' changes are summed over RCCL.
// These kernels are the arithmetic around that all-reduce, fused into one pass over the tables per merge:
// written as torch elementwise chains they moved ~3x the bytes (7 passes per table), and at a few hundred merges
// per pass of a 1 GB table pair that costs as much as the training itself.  One wavefront per row, float2 (or
// wider) per lane; pure streaming, bound by HBM bandwidth: 28 B per element for the snapshot pass.
#include "n2v_common.h"

namespace {

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round to nearest even (what torch's .to(bfloat16) does)
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
template <bool BF16> __device__ __forceinline__ float wire_load(const void* p, int64_t i) {
    return BF16 ? bf16_to_f32(reinterpret_cast<const uint16_t*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
template <bool BF16> __device__ __forceinline__ void wire_store(void* p, int64_t i, float v) {
    if (BF16) reinterpret_cast<uint16_t*>(p)[i] = f32_to_bf16(v);
    else reinterpret_cast<float*>(p)[i] = v;
}

template <bool BF16>
__global__ void __launch_bounds__(256)
merge_snapshot_kernel(float* __restrict__ x, float* __restrict__ xs, float* __restrict__ base, int64_t n_rows, int stride,
                      const float* __restrict__ w, const int32_t* __restrict__ hot_pos, const void* __restrict__ sum_prev,
                      void* __restrict__ cold_wire, void* __restrict__ hot_wire) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    const int hp = hot_pos ? hot_pos[r] : -1;
    const float wr = w[r];
    const int64_t o = r * stride;
    for (int c = lane; c < stride; c += 64) {
        const float d = x[o + c] - xs[o + c];
        if (hp >= 0) {
            wire_store<BF16>(hot_wire, (int64_t)hp * stride + c, d);
            if (cold_wire) wire_store<BF16>(cold_wire, o + c, 0.f);
        } else {
            float b = base[o + c];
            if (sum_prev) { b += wr * wire_load<BF16>(sum_prev, o + c); base[o + c] = b; }
            wire_store<BF16>(cold_wire, o + c, d);
            const float nx = b + d;      // the rank keeps its own not-yet-merged change
            x[o + c] = nx;
            xs[o + c] = nx;
        }
    }
}

template <bool BF16>
__global__ void __launch_bounds__(256)
merge_hot_apply_kernel(float* __restrict__ x, float* __restrict__ xs, float* __restrict__ base, int stride,
                       const float* __restrict__ w, const int64_t* __restrict__ hot_rows, int64_t n_hot,
                       const void* __restrict__ hot_sum) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (j >= n_hot) return;
    const int64_t r = hot_rows[j];
    const float wr = w[r];
    const int64_t o = r * stride;
    for (int c = lane; c < stride; c += 64) {
        const float b = base[o + c] + wr * wire_load<BF16>(hot_sum, j * stride + c);
        base[o + c] = b;
        x[o + c] = b;
        xs[o + c] = b;
    }
}

template <bool BF16>
__global__ void __launch_bounds__(256)
merge_flush_kernel(float* __restrict__ x, float* __restrict__ xs, float* __restrict__ base, int64_t n_rows, int stride,
                   const float* __restrict__ w, const int32_t* __restrict__ hot_pos, const void* __restrict__ sum_last) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    const bool cold = !hot_pos || hot_pos[r] < 0;
    const float wr = w[r];
    const int64_t o = r * stride;
    for (int c = lane; c < stride; c += 64) {
        float b = base[o + c];
        if (cold && sum_last) { b += wr * wire_load<BF16>(sum_last, o + c); base[o + c] = b; }
        x[o + c] = b;
        xs[o + c] = b;
    }
}

// wire[j] = x[rows[j]] - base[rows[j]]: the change of a row LIST since its last merge (tiered pure-sum merges)
template <bool BF16>
__global__ void __launch_bounds__(256)
merge_pack_rows_kernel(const float* __restrict__ x, const float* __restrict__ base, int stride, const int64_t* __restrict__ rows,
                       int64_t n_list, void* __restrict__ wire) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (j >= n_list) return;
    const int64_t o = rows[j] * stride;
    for (int c = lane; c < stride; c += 64) wire_store<BF16>(wire, j * stride + c, x[o + c] - base[o + c]);
}

// Tiered sums, all tables of a merge in ONE launch (the hub tiers are merged ~15 000 times per pass at 8 GPUs and hold
// a few thousand rows: launch count, not bytes, is what they cost).  Wire row j of the launch belongs to table t
// (first[t] <= j < first[t+1]), list position j - first[t]; rows == NULL: the list is every row in order.
struct TsumTabs {
    float* x[N2V_TSUM_MAX_TABLES];
    float* base[N2V_TSUM_MAX_TABLES];
    const int64_t* rows[N2V_TSUM_MAX_TABLES];
    int64_t first[N2V_TSUM_MAX_TABLES + 1];
    int n_tabs;
    int stride;
};

template <bool BF16, bool APPLY>
__global__ void __launch_bounds__(256) tsum_kernel(TsumTabs a, void* __restrict__ wire) {
    const int lane = threadIdx.x & 63;
    const int64_t j = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (j >= a.first[a.n_tabs]) return;
    // constant indices only: a run-time index into the argument struct would send it through scratch memory
    float* __restrict__ x = a.x[0];
    float* __restrict__ base = a.base[0];
    const int64_t* __restrict__ rows = a.rows[0];
    int64_t first = 0;
#pragma unroll
    for (int t = 1; t < N2V_TSUM_MAX_TABLES; ++t)
        if (t < a.n_tabs && j >= a.first[t]) { x = a.x[t]; base = a.base[t]; rows = a.rows[t]; first = a.first[t]; }
    const int64_t k = j - first;
    const int64_t o = (rows ? rows[k] : k) * a.stride;
    for (int c = lane; c < a.stride; c += 64) {
        if (APPLY) {
            const float b = base[o + c] + wire_load<BF16>(wire, j * a.stride + c);
            base[o + c] = b;
            x[o + c] = b;
        } else {
            wire_store<BF16>(wire, j * a.stride + c, x[o + c] - base[o + c]);
        }
    }
}

int tsum_launch(const char* what, const n2v_tsum_table* tabs, int32_t n_tabs, int32_t stride, void* wire, int32_t wire_bf16,
                bool apply, void* stream) {
    if (n_tabs < 0 || n_tabs > N2V_TSUM_MAX_TABLES || stride < 1) return n2v::fail(N2V_ERR_INVALID, "%s: bad sizes", what);
    if (n_tabs && !tabs) return n2v::fail(N2V_ERR_INVALID, "%s: null pointer", what);
    TsumTabs a{};
    a.n_tabs = n_tabs;
    a.stride = stride;
    int64_t total = 0;
    for (int t = 0; t < n_tabs; ++t) {
        if (tabs[t].n_rows < 0) return n2v::fail(N2V_ERR_INVALID, "%s: negative row count", what);
        if (tabs[t].n_rows && (!tabs[t].table || !tabs[t].base)) return n2v::fail(N2V_ERR_INVALID, "%s: null table", what);
        a.x[t] = tabs[t].table;
        a.base[t] = tabs[t].base;
        a.rows[t] = tabs[t].rows;
        a.first[t] = total;
        total += tabs[t].n_rows;
    }
    for (int t = n_tabs; t <= N2V_TSUM_MAX_TABLES; ++t) a.first[t] = total;
    if (total == 0) return N2V_OK;
    if (!wire) return n2v::fail(N2V_ERR_INVALID, "%s: null wire buffer", what);
    const int64_t blocks = (total + 3) / 4;
    if (blocks > 0x7fffffff) return n2v::fail(N2V_ERR_INVALID, "%s: too many rows", what);
    const dim3 g((unsigned)blocks), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (apply) {
        if (wire_bf16) hipLaunchKernelGGL((tsum_kernel<true, true>), g, b, 0, s, a, wire);
        else hipLaunchKernelGGL((tsum_kernel<false, true>), g, b, 0, s, a, wire);
    } else {
        if (wire_bf16) hipLaunchKernelGGL((tsum_kernel<true, false>), g, b, 0, s, a, wire);
        else hipLaunchKernelGGL((tsum_kernel<false, false>), g, b, 0, s, a, wire);
    }
    return n2v::check_launch(what);
}

int rows_grid(int64_t n, unsigned* blocks) {
    const int64_t b = (n + 3) / 4;
    if (b > 0x7fffffff) return -1;
    *blocks = (unsigned)b;
    return 0;
}

}  // namespace

extern "C" int n2v_merge_snapshot(float* x, float* xs, float* base, int64_t n_rows, int32_t stride, const float* w,
                                  const int32_t* hot_pos, const void* cold_sum_prev, void* cold_wire, void* hot_wire,
                                  int32_t wire_bf16, void* stream) {
    if (n_rows < 0 || stride < 1) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_snapshot: bad sizes");
    if (n_rows == 0) return N2V_OK;
    if (!x || !xs || !base || !w) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_snapshot: null pointer");
    if (!cold_wire && !hot_pos) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_snapshot: cold rows need a wire buffer");
    if (hot_pos && !hot_wire) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_snapshot: hot rows need a wire buffer");
    unsigned blocks;
    if (rows_grid(n_rows, &blocks)) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_snapshot: too many rows");
    if (wire_bf16) hipLaunchKernelGGL((merge_snapshot_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, xs, base,
                                      n_rows, (int)stride, w, hot_pos, cold_sum_prev, cold_wire, hot_wire);
    else hipLaunchKernelGGL((merge_snapshot_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, xs, base, n_rows,
                            (int)stride, w, hot_pos, cold_sum_prev, cold_wire, hot_wire);
    return n2v::check_launch("n2v_merge_snapshot");
}

extern "C" int n2v_merge_hot_apply(float* x, float* xs, float* base, int32_t stride, const float* w,
                                   const int64_t* hot_rows, int64_t n_hot, const void* hot_sum, int32_t wire_bf16,
                                   void* stream) {
    if (n_hot < 0 || stride < 1) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_hot_apply: bad sizes");
    if (n_hot == 0) return N2V_OK;
    if (!x || !xs || !base || !w || !hot_rows || !hot_sum) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_hot_apply: null pointer");
    unsigned blocks;
    if (rows_grid(n_hot, &blocks)) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_hot_apply: too many rows");
    if (wire_bf16) hipLaunchKernelGGL((merge_hot_apply_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, xs, base,
                                      (int)stride, w, hot_rows, n_hot, hot_sum);
    else hipLaunchKernelGGL((merge_hot_apply_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, xs, base,
                            (int)stride, w, hot_rows, n_hot, hot_sum);
    return n2v::check_launch("n2v_merge_hot_apply");
}

extern "C" int n2v_merge_flush(float* x, float* xs, float* base, int64_t n_rows, int32_t stride, const float* w,
                               const int32_t* hot_pos, const void* cold_sum_last, int32_t wire_bf16, void* stream) {
    if (n_rows < 0 || stride < 1) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_flush: bad sizes");
    if (n_rows == 0) return N2V_OK;
    if (!x || !xs || !base || !w) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_flush: null pointer");
    unsigned blocks;
    if (rows_grid(n_rows, &blocks)) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_flush: too many rows");
    if (wire_bf16) hipLaunchKernelGGL((merge_flush_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, xs, base, n_rows,
                                      (int)stride, w, hot_pos, cold_sum_last);
    else hipLaunchKernelGGL((merge_flush_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, xs, base, n_rows,
                            (int)stride, w, hot_pos, cold_sum_last);
    return n2v::check_launch("n2v_merge_flush");
}

extern "C" int n2v_merge_pack_rows(const float* x, const float* base, int32_t stride, const int64_t* rows, int64_t n_list,
                                   void* wire, int32_t wire_bf16, void* stream) {
    if (n_list < 0 || stride < 1) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_pack_rows: bad sizes");
    if (n_list == 0) return N2V_OK;
    if (!x || !base || !rows || !wire) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_pack_rows: null pointer");
    unsigned blocks;
    if (rows_grid(n_list, &blocks)) return n2v::fail(N2V_ERR_INVALID, "n2v_merge_pack_rows: too many rows");
    if (wire_bf16) hipLaunchKernelGGL((merge_pack_rows_kernel<true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, base,
                                      (int)stride, rows, n_list, wire);
    else hipLaunchKernelGGL((merge_pack_rows_kernel<false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, base, (int)stride,
                            rows, n_list, wire);
    return n2v::check_launch("n2v_merge_pack_rows");
}

extern "C" int n2v_tsum_pack(const n2v_tsum_table* tabs, int32_t n_tabs, int32_t stride, void* wire, int32_t wire_bf16,
                             void* stream) {
    return tsum_launch("n2v_tsum_pack", tabs, n_tabs, stride, wire, wire_bf16, false, stream);
}

extern "C" int n2v_tsum_apply(const n2v_tsum_table* tabs, int32_t n_tabs, int32_t stride, const void* wire,
                              int32_t wire_bf16, void* stream) {
    return tsum_launch("n2v_tsum_apply", tabs, n_tabs, stride, const_cast<void*>(wire), wire_bf16, true, stream);
}
