// All-pairs similarity + selection — gfx950 (MI355X).  C-ABI: include/n2v_sim.h.
//
// Reference: src/main_link.py:62-170 (link_prediction: score every candidate pair, keep the k best) and
// :351-453 (user x user similarity and the per-user selection of edges to add).  Both are Python double
// loops over gensim's `similarity` there.  Here one 64x64-tile kernel (256 threads, 4x4 scores per thread,
// operands staged k-major through LDS, fp32 FMA) produces the scores of a row block against all columns and
//   * EPI_SCAN : appends the scores above a running threshold to a candidate buffer (global top-k), or
//   * EPI_BLOCK: writes the row block once; one workgroup per row then selects by threshold or by an exact
//                radix select of the k-th largest score, emitting in column (= list) order.
// A genuine dense contraction in fp32 (the reference's float32 dot): the dot-product forms ("cos", "pearson") run on
// the matrix cores (sim_mfma_kernel, v_mfma_f32_32x32x2_f32: fp32 in and out, so the top-k SETS can be compared
// with a CPU restatement); the Jensen-Shannon form is not a contraction and stays on the vector tile kernel.
#include "n2v_common.h"
#include "n2v_sim.h"

#include <cstdlib>

namespace {

constexpr int TILE = 64;   // scores per tile edge
constexpr int KC = 32;     // k-chunk staged per barrier
constexpr int LDT = 68;    // LDS row pitch (floats): float4-aligned, store conflicts 2-way at most

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------- prepare
__global__ void __launch_bounds__(256)
sim_prepare_kernel(const float* __restrict__ vec, int stride, int dim, const int64_t* __restrict__ rows, int64_t n_rows,
                   int method, float* __restrict__ out, int dpad) {
    const int lane = threadIdx.x & 63;
    const int64_t r = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (r >= n_rows) return;
    const float* x = vec + (rows ? rows[r] : r) * (int64_t)stride;
    float* o = out + r * (int64_t)dpad;
    float s = 0.f, s2 = 0.f;
    for (int k = lane; k < dim; k += 64) { const float v = x[k]; s += v; s2 += v * v; }
    s = wave_sum(s);
    if (method == N2V_SIM_COS) {
        s2 = wave_sum(s2);
        // gensim unitvec: scal(1/nrm2, x) if nrm2 > 0 else x — a zero row stays zero (similarity 0), not NaN
        const float inv = s2 > 0.f ? 1.0f / sqrtf(s2) : 1.0f;
        for (int k = lane; k < dpad; k += 64) o[k] = k < dim ? x[k] * inv : 0.f;
    } else if (method == N2V_SIM_PEARSON) {
        const float mean = s / (float)dim;                        // scipy pearsonr: xm = x - mean; xm / norm(xm)
        float c2 = 0.f;
        for (int k = lane; k < dim; k += 64) { const float v = x[k] - mean; c2 += v * v; }
        c2 = wave_sum(c2);
        const float nrm = sqrtf(c2);
        for (int k = lane; k < dpad; k += 64) o[k] = k < dim ? (x[k] - mean) / nrm : 0.f;
    } else {                                                      // js(): p / p.sum()
        for (int k = lane; k < dpad; k += 64) o[k] = k < dim ? x[k] / s : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------- tile
// scipy.special.rel_entr: x log(x/y) for x, y > 0; 0 for x == 0, y >= 0; +inf otherwise.
__device__ __forceinline__ float rel_entr(float x, float y) {
    if (x > 0.f && y > 0.f) return x * logf(x / y);
    if (x == 0.f && y >= 0.f) return 0.f;
    return (x != x || y != y) ? x + y : __builtin_inff();
}

struct TileArgs {
    const float* A; const float* B;
    int64_t row_begin, row_end, n_cols;
    int dpad;
    // EPI_BLOCK
    float* out; int64_t ld; int64_t zero_diag_off;
    // EPI_SCAN
    int upper; const float* tau; const int64_t* excl; int64_t n_excl;
    float* cand_score; int32_t* cand_row; int32_t* cand_col; int64_t capacity; int64_t* counter;
};

__device__ __forceinline__ bool key_in(const int64_t* __restrict__ keys, int64_t n, int64_t key) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo < n && keys[lo] == key;
}

template <bool JSD, bool SCAN>
__global__ void __launch_bounds__(256) sim_tile_kernel(TileArgs a) {
    __shared__ __attribute__((aligned(16))) float As[KC][LDT];
    __shared__ __attribute__((aligned(16))) float Bs[KC][LDT];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int64_t r0 = a.row_begin + (int64_t)blockIdx.y * TILE, c0 = (int64_t)blockIdx.x * TILE;
    if (SCAN && a.upper && c0 + TILE - 1 <= r0) return;        // whole tile on or below the diagonal
    const int lrow = tid >> 2, kq = (tid & 3) * 8;
    const int64_t ar = r0 + lrow, br = c0 + lrow;
    const float* ap = a.A + ar * a.dpad + kq;
    const float* bp = a.B + br * a.dpad + kq;
    const bool a_ok = ar < a.row_end, b_ok = br < a.n_cols;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < a.dpad; k0 += KC) {
        float4 a0 = {0, 0, 0, 0}, a1 = a0, b0 = a0, b1 = a0;
        if (a_ok) { a0 = *reinterpret_cast<const float4*>(ap + k0); a1 = *reinterpret_cast<const float4*>(ap + k0 + 4); }
        if (b_ok) { b0 = *reinterpret_cast<const float4*>(bp + k0); b1 = *reinterpret_cast<const float4*>(bp + k0 + 4); }
        __syncthreads();
        As[kq + 0][lrow] = a0.x; As[kq + 1][lrow] = a0.y; As[kq + 2][lrow] = a0.z; As[kq + 3][lrow] = a0.w;
        As[kq + 4][lrow] = a1.x; As[kq + 5][lrow] = a1.y; As[kq + 6][lrow] = a1.z; As[kq + 7][lrow] = a1.w;
        Bs[kq + 0][lrow] = b0.x; Bs[kq + 1][lrow] = b0.y; Bs[kq + 2][lrow] = b0.z; Bs[kq + 3][lrow] = b0.w;
        Bs[kq + 4][lrow] = b1.x; Bs[kq + 5][lrow] = b1.y; Bs[kq + 6][lrow] = b1.z; Bs[kq + 7][lrow] = b1.w;
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < KC; ++k) {
            const float4 av = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
            const float4 bv = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
            const float ar4[4] = {av.x, av.y, av.z, av.w}, bc4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (JSD) {
                        const float m = (ar4[i] + bc4[j]) * 0.5f;           // js(): m = (p + q) / 2
                        acc[i][j] += (rel_entr(ar4[i], m) + rel_entr(bc4[j], m)) * 0.5f;
                    } else {
                        acc[i][j] = fmaf(ar4[i], bc4[j], acc[i][j]);
                    }
                }
        }
    }
    if (SCAN) {
        const float tau = *a.tau;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = r0 + ty * 4 + i;
            if (r >= a.row_end) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t c = c0 + tx * 4 + j;
                const float s = acc[i][j];
                if (c >= a.n_cols || !(s > tau) || (a.upper && c <= r)) continue;
                if (a.n_excl > 0 && key_in(a.excl, a.n_excl, r * a.n_cols + c)) continue;   // train edge (:74,:84)
                const int64_t idx = (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(a.counter), 1ull);
                if (idx < a.capacity) { a.cand_score[idx] = s; a.cand_row[idx] = (int32_t)r; a.cand_col[idx] = (int32_t)c; }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = r0 + ty * 4 + i;
            if (r >= a.row_end) continue;
            float* o = a.out + (r - a.row_begin) * a.ld + c0 + tx * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t c = c0 + tx * 4 + j;
                if (c >= a.n_cols) continue;
                o[j] = (a.zero_diag_off >= 0 && c == r + a.zero_diag_off) ? 0.f : acc[i][j];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- MFMA tile (dot)
// 128x128 scores per workgroup, 64x64 per wavefront as 2x2 blocks of v_mfma_f32_32x32x2_f32 (fp32 in, fp32
// accumulate: the reference's float32 dot, no reduced precision).  Operands staged k-major through LDS like the
// vector kernel; per k-pair a wave issues 4 LDS reads for 4 MFMAs (8 192 FMAs), against 8 LDS floats per 16 FMAs
// in the vector form — the matrix pipe is what a dense fp32 contraction is for on this chip.  Operand layout of
// the instruction: lane l supplies A[m = l % 32][k = l / 32] and B[k = l / 32][n = l % 32]; accumulator register v
// of lane l holds D[8 * (v / 4) + 4 * (l / 32) + (v % 4)][l % 32].
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int MT = 128;    // tile edge
constexpr int MKC = 16;    // k-chunk per barrier
constexpr int MLD = 132;   // LDS pitch: the two k-halves of a wave's store land 32 banks apart

template <bool SCAN>
__global__ void __launch_bounds__(256) sim_mfma_kernel(TileArgs a) {
    __shared__ __attribute__((aligned(16))) float As[MKC][MLD];
    __shared__ __attribute__((aligned(16))) float Bs[MKC][MLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1;
    const int64_t r0 = a.row_begin + (int64_t)blockIdx.y * MT, c0 = (int64_t)blockIdx.x * MT;
    if (SCAN && a.upper && c0 + MT - 1 <= r0) return;
    const int lrow = tid >> 1, kq = (tid & 1) * 8;
    const int64_t ar = r0 + lrow, br = c0 + lrow;
    const float* ap = a.A + ar * a.dpad + kq;
    const float* bp = a.B + br * a.dpad + kq;
    const bool a_ok = ar < a.row_end, b_ok = br < a.n_cols;
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    const float4 z4 = {0, 0, 0, 0};
    float4 a0 = z4, a1 = z4, b0 = z4, b1 = z4;
    if (a_ok) { a0 = *reinterpret_cast<const float4*>(ap); a1 = *reinterpret_cast<const float4*>(ap + 4); }
    if (b_ok) { b0 = *reinterpret_cast<const float4*>(bp); b1 = *reinterpret_cast<const float4*>(bp + 4); }
    const int kh = lane >> 5, m = lane & 31;
    for (int k0 = 0; k0 < a.dpad; k0 += MKC) {
        __syncthreads();
        As[kq + 0][lrow] = a0.x; As[kq + 1][lrow] = a0.y; As[kq + 2][lrow] = a0.z; As[kq + 3][lrow] = a0.w;
        As[kq + 4][lrow] = a1.x; As[kq + 5][lrow] = a1.y; As[kq + 6][lrow] = a1.z; As[kq + 7][lrow] = a1.w;
        Bs[kq + 0][lrow] = b0.x; Bs[kq + 1][lrow] = b0.y; Bs[kq + 2][lrow] = b0.z; Bs[kq + 3][lrow] = b0.w;
        Bs[kq + 4][lrow] = b1.x; Bs[kq + 5][lrow] = b1.y; Bs[kq + 6][lrow] = b1.z; Bs[kq + 7][lrow] = b1.w;
        __syncthreads();
        if (k0 + MKC < a.dpad) {      // next chunk's rows travel while this one is multiplied
            a0 = a1 = b0 = b1 = z4;
            if (a_ok) { a0 = *reinterpret_cast<const float4*>(ap + k0 + MKC); a1 = *reinterpret_cast<const float4*>(ap + k0 + MKC + 4); }
            if (b_ok) { b0 = *reinterpret_cast<const float4*>(bp + k0 + MKC); b1 = *reinterpret_cast<const float4*>(bp + k0 + MKC + 4); }
        }
#pragma unroll
        for (int kk = 0; kk < MKC; kk += 2) {
            const float av0 = As[kk + kh][wr * 64 + m], av1 = As[kk + kh][wr * 64 + 32 + m];
            const float bv0 = Bs[kk + kh][wc * 64 + m], bv1 = Bs[kk + kh][wc * 64 + 32 + m];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1, acc[1][1], 0, 0, 0);
        }
    }
    const float tau = SCAN ? *a.tau : 0.f;
#pragma unroll
    for (int bi = 0; bi < 2; ++bi)
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int64_t r = r0 + wr * 64 + bi * 32 + 8 * (v >> 2) + 4 * kh + (v & 3);
                const int64_t c = c0 + wc * 64 + bj * 32 + m;
                if (r >= a.row_end || c >= a.n_cols) continue;
                const float s = acc[bi][bj][v];
                if (SCAN) {
                    if (!(s > tau) || (a.upper && c <= r)) continue;
                    if (a.n_excl > 0 && key_in(a.excl, a.n_excl, r * a.n_cols + c)) continue;   // train edge (:74,:84)
                    const int64_t idx = (int64_t)atomicAdd(reinterpret_cast<unsigned long long*>(a.counter), 1ull);
                    if (idx < a.capacity) { a.cand_score[idx] = s; a.cand_row[idx] = (int32_t)r; a.cand_col[idx] = (int32_t)c; }
                } else {
                    a.out[(r - a.row_begin) * a.ld + c] = (a.zero_diag_off >= 0 && c == r + a.zero_diag_off) ? 0.f : s;
                }
            }
}

// ---------------------------------------------------------------------------------------------- row selection
// Ordered emission helper: every thread of the 256-thread workgroup passes `flag`; returns the thread's
// position among the flagged threads of this call plus *base (shared running total, advanced by the call).
__device__ __forceinline__ int64_t ordered_slot(bool flag, int64_t* base, int* wave_tot) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wv] = __popcll(m);
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int t = wave_tot[w]; if (w < wv) off += t; tot += t; }
    const int64_t pos = *base + off + before;
    __syncthreads();
    if (threadIdx.x == 0) *base += tot;
    __syncthreads();
    return pos;
}

__global__ void __launch_bounds__(256)
rows_count_kernel(const float* __restrict__ scores, int64_t n_cols, int64_t ld, float thre, int64_t* __restrict__ counts) {
    __shared__ int part[4];
    const float* s = scores + (int64_t)blockIdx.x * ld;
    int cnt = 0;
    for (int64_t c = threadIdx.x; c < n_cols; c += 256) cnt += s[c] > thre;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = (int64_t)part[0] + part[1] + part[2] + part[3];
}

__global__ void __launch_bounds__(256)
rows_fill_kernel(const float* __restrict__ scores, int64_t n_cols, int64_t ld, float thre, const int64_t* __restrict__ out_off,
                 int32_t* __restrict__ cols, float* __restrict__ vals) {
    __shared__ int64_t base;
    __shared__ int wave_tot[4];
    const float* s = scores + (int64_t)blockIdx.x * ld;
    if (threadIdx.x == 0) base = out_off[blockIdx.x];
    __syncthreads();
    for (int64_t c0 = 0; c0 < n_cols; c0 += 256) {
        const int64_t c = c0 + threadIdx.x;
        const float v = c < n_cols ? s[c] : 0.f;
        const bool f = c < n_cols && v > thre;
        const int64_t pos = ordered_slot(f, &base, wave_tot);
        if (f) { cols[pos] = (int32_t)c; vals[pos] = v; }
    }
}

// order-preserving key: larger float <=> larger key; NaN lowest
__device__ __forceinline__ uint32_t order_key(float v) {
    if (v != v) return 0u;
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ void __launch_bounds__(256)
rows_topk_kernel(const float* __restrict__ scores, int64_t n_cols, int64_t ld, int k, int32_t* __restrict__ cols,
                 float* __restrict__ vals) {
    __shared__ int hist[256];
    __shared__ uint32_t sel_prefix;
    __shared__ int sel_remaining;
    __shared__ int64_t base;
    __shared__ int wave_tot[4];
    __shared__ int eq_base;
    const float* s = scores + (int64_t)blockIdx.x * ld;
    if (threadIdx.x == 0) { sel_prefix = 0u; sel_remaining = k; }
    // radix select of the k-th largest key, 8 bits per pass from the top
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t prefix = sel_prefix;
        const uint32_t himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int64_t c = threadIdx.x; c < n_cols; c += 256) {
            const uint32_t key = order_key(s[c]);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int rem = sel_remaining, b = 255;
            for (; b > 0; --b) { if (hist[b] >= rem) break; rem -= hist[b]; }
            sel_prefix = prefix | ((uint32_t)b << shift);
            sel_remaining = rem;                                // rank of the k-th inside bin b (1-based)
        }
        __syncthreads();
    }
    const uint32_t T = sel_prefix;
    const int need_eq = sel_remaining;                          // how many keys == T belong to the top k
    if (threadIdx.x == 0) { base = (int64_t)blockIdx.x * k; eq_base = 0; }
    __syncthreads();
    for (int64_t c0 = 0; c0 < n_cols; c0 += 256) {
        const int64_t c = c0 + threadIdx.x;
        const float v = c < n_cols ? s[c] : 0.f;
        const uint32_t key = c < n_cols ? order_key(v) : 0u;
        const bool gt = c < n_cols && key > T;
        bool eq = c < n_cols && key == T;
        // ties of the k-th value: the first need_eq of them in column order (stable sort keeps list order)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const unsigned long long m = __ballot(eq);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int t = wave_tot[w]; if (w < wv) off += t; tot += t; }
        const int eq_rank = eq_base + off + before;
        __syncthreads();
        if (threadIdx.x == 0) eq_base += tot;
        eq = eq && eq_rank < need_eq;
        const bool f = gt || eq;
        const int64_t pos = ordered_slot(f, &base, wave_tot);
        if (f) { cols[pos] = (int32_t)c; vals[pos] = v; }
    }
}

}  // namespace

// ================================================================================================== C-ABI
extern "C" int n2v_sim_prepare(const float* vec, int32_t stride, int32_t dim, const int64_t* rows, int64_t n_rows,
                               int32_t method, float* out, int32_t dpad, void* stream) {
    if (n_rows < 0 || dim < 1 || stride < dim || dpad < dim || (dpad % KC) != 0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_sim_prepare: bad sizes (dim %d stride %d dpad %d)", (int)dim, (int)stride, (int)dpad);
    if (method < N2V_SIM_COS || method > N2V_SIM_JSD) return n2v::fail(N2V_ERR_INVALID, "n2v_sim_prepare: method %d", (int)method);
    if (n_rows == 0) return N2V_OK;
    if (!vec || !out) return n2v::fail(N2V_ERR_INVALID, "n2v_sim_prepare: null pointer");
    const int64_t blocks = (n_rows + 3) / 4;
    if (blocks > 0x7fffffff) return n2v::fail(N2V_ERR_INVALID, "n2v_sim_prepare: too many rows");
    hipLaunchKernelGGL(sim_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, vec, (int)stride,
                       (int)dim, rows, n_rows, (int)method, out, (int)dpad);
    return n2v::check_launch("n2v_sim_prepare");
}

// N2V_SIM_VECTOR=1 runs the dot product on the vector-FMA tile kernel instead of the MFMA one (A/B in tools/sim_probe.py)
static bool vector_fma_forced() { const char* e = getenv("N2V_SIM_VECTOR"); return e && e[0] == '1'; }

static int tile_args_ok(const char* who, const float* A, const float* B, int64_t row_begin, int64_t n_rows, int64_t n_cols,
                        int32_t dpad) {
    if (row_begin < 0 || n_rows < 0 || n_cols < 0 || dpad < KC || (dpad % KC) != 0)
        return n2v::fail(N2V_ERR_INVALID, "%s: bad sizes", who);
    if (n_rows > 0 && n_cols > 0 && (!A || !B)) return n2v::fail(N2V_ERR_INVALID, "%s: null pointer", who);
    if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0) return n2v::fail(N2V_ERR_INVALID, "%s: operands not 16-byte aligned", who);
    if ((n_rows + TILE - 1) / TILE > 65535) return n2v::fail(N2V_ERR_INVALID, "%s: more than 65535 row tiles in one call", who);
    if (n_cols > (int64_t)0x7fffffff) return n2v::fail(N2V_ERR_INVALID, "%s: too many columns", who);
    return N2V_OK;
}

extern "C" int n2v_sim_block(const float* A, int64_t row_begin, int64_t n_rows, const float* B, int64_t n_cols,
                             int32_t dpad, int32_t method, int64_t zero_diag_off, float* out, int64_t ld, void* stream) {
    const int rc = tile_args_ok("n2v_sim_block", A, B, row_begin, n_rows, n_cols, dpad);
    if (rc != N2V_OK) return rc;
    if (n_rows == 0 || n_cols == 0) return N2V_OK;
    if (!out || ld < n_cols) return n2v::fail(N2V_ERR_INVALID, "n2v_sim_block: bad output");
    TileArgs a{};
    a.A = A; a.B = B; a.row_begin = row_begin; a.row_end = row_begin + n_rows; a.n_cols = n_cols; a.dpad = dpad;
    a.out = out; a.ld = ld; a.zero_diag_off = zero_diag_off;
    const dim3 grid((unsigned)((n_cols + TILE - 1) / TILE), (unsigned)((n_rows + TILE - 1) / TILE));
    const dim3 mgrid((unsigned)((n_cols + MT - 1) / MT), (unsigned)((n_rows + MT - 1) / MT));
    if (method == N2V_SIM_JSD) hipLaunchKernelGGL((sim_tile_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (vector_fma_forced()) hipLaunchKernelGGL((sim_tile_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((sim_mfma_kernel<false>), mgrid, dim3(256), 0, (hipStream_t)stream, a);
    return n2v::check_launch("n2v_sim_block");
}

extern "C" int n2v_sim_topk_scan(const float* A, int64_t row_begin, int64_t row_end, const float* B, int64_t n_cols,
                                 int32_t dpad, int32_t method, int32_t upper_triangle, const float* tau,
                                 const int64_t* excl_keys, int64_t n_excl, float* cand_score, int32_t* cand_row,
                                 int32_t* cand_col, int64_t capacity, int64_t* counter, void* stream) {
    const int rc = tile_args_ok("n2v_sim_topk_scan", A, B, row_begin, row_end - row_begin, n_cols, dpad);
    if (rc != N2V_OK) return rc;
    if (row_end <= row_begin || n_cols == 0) return N2V_OK;
    if (!tau || !cand_score || !cand_row || !cand_col || !counter || capacity < 1 || n_excl < 0 || (n_excl > 0 && !excl_keys))
        return n2v::fail(N2V_ERR_INVALID, "n2v_sim_topk_scan: null pointer or bad capacity");
    if (row_end > (int64_t)0x7fffffff) return n2v::fail(N2V_ERR_INVALID, "n2v_sim_topk_scan: too many rows");
    TileArgs a{};
    a.A = A; a.B = B; a.row_begin = row_begin; a.row_end = row_end; a.n_cols = n_cols; a.dpad = dpad;
    a.upper = upper_triangle ? 1 : 0; a.tau = tau; a.excl = excl_keys; a.n_excl = n_excl;
    a.cand_score = cand_score; a.cand_row = cand_row; a.cand_col = cand_col; a.capacity = capacity; a.counter = counter;
    const dim3 grid((unsigned)((n_cols + TILE - 1) / TILE), (unsigned)((row_end - row_begin + TILE - 1) / TILE));
    const dim3 mgrid((unsigned)((n_cols + MT - 1) / MT), (unsigned)((row_end - row_begin + MT - 1) / MT));
    if (method == N2V_SIM_JSD) hipLaunchKernelGGL((sim_tile_kernel<true, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (vector_fma_forced()) hipLaunchKernelGGL((sim_tile_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((sim_mfma_kernel<true>), mgrid, dim3(256), 0, (hipStream_t)stream, a);
    return n2v::check_launch("n2v_sim_topk_scan");
}

static int rows_args_ok(const char* who, const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld) {
    if (n_rows < 0 || n_cols < 0 || ld < n_cols || n_rows > 0x7fffffff) return n2v::fail(N2V_ERR_INVALID, "%s: bad sizes", who);
    if (n_rows > 0 && !scores) return n2v::fail(N2V_ERR_INVALID, "%s: null pointer", who);
    return N2V_OK;
}

extern "C" int n2v_sim_rows_count(const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, float thre,
                                  int64_t* counts, void* stream) {
    const int rc = rows_args_ok("n2v_sim_rows_count", scores, n_rows, n_cols, ld);
    if (rc != N2V_OK) return rc;
    if (n_rows == 0) return N2V_OK;
    if (!counts) return n2v::fail(N2V_ERR_INVALID, "n2v_sim_rows_count: null pointer");
    hipLaunchKernelGGL(rows_count_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, scores, n_cols, ld, thre, counts);
    return n2v::check_launch("n2v_sim_rows_count");
}

extern "C" int n2v_sim_rows_fill(const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, float thre,
                                 const int64_t* out_off, int32_t* cols, float* vals, void* stream) {
    const int rc = rows_args_ok("n2v_sim_rows_fill", scores, n_rows, n_cols, ld);
    if (rc != N2V_OK) return rc;
    if (n_rows == 0) return N2V_OK;
    if (!out_off || !cols || !vals) return n2v::fail(N2V_ERR_INVALID, "n2v_sim_rows_fill: null pointer");
    hipLaunchKernelGGL(rows_fill_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, scores, n_cols, ld, thre,
                       out_off, cols, vals);
    return n2v::check_launch("n2v_sim_rows_fill");
}

extern "C" int n2v_sim_rows_topk(const float* scores, int64_t n_rows, int64_t n_cols, int64_t ld, int32_t k,
                                 int32_t* cols, float* vals, void* stream) {
    const int rc = rows_args_ok("n2v_sim_rows_topk", scores, n_rows, n_cols, ld);
    if (rc != N2V_OK) return rc;
    if (k < 0 || k > n_cols) return n2v::fail(N2V_ERR_INVALID, "n2v_sim_rows_topk: k %d outside [0, %lld]", (int)k, (long long)n_cols);
    if (n_rows == 0 || k == 0) return N2V_OK;
    if (!cols || !vals) return n2v::fail(N2V_ERR_INVALID, "n2v_sim_rows_topk: null pointer");
    hipLaunchKernelGGL(rows_topk_kernel, dim3((unsigned)n_rows), dim3(256), 0, (hipStream_t)stream, scores, n_cols, ld, (int)k, cols, vals);
    return n2v::check_launch("n2v_sim_rows_topk");
}
