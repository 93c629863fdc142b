// BiNE path (bipartite network embedding) — gfx950 (MI355X) kernels and C-ABI (include/n2v_bine.h).
//
// Replaces the reference's src/bine_train.py (train / skip_gram / KL_divergence /
// init_embedding_vectors), src/bine_graph_utils.py (calculate_centrality, the projected-graph
// restart walks, get_context_and_negatives), src/bine_graph.py (random_walk_restart_for_large_
// bipartite_graph, build_deepwalk_corpus_random) and the role of src/bine_lsh.py.
//
// Mapping to the machine
//  * the bipartite graph is one symmetric CSR over users+items; the projections A*A^T / A^T*A
//    are never built — a step proposes a two-hop path and keeps it only when its middle vertex is
//    the first common neighbour of its ends, which is the reference's uniform draw over the
//    DISTINCT two-hop vertices (n2v_bine.h);
//  * one wavefront per walk: the proposal is scalar work shared by the wave, the search for an
//    earlier common neighbour is done by the 64 lanes together (binary searches + ballot, early exit);
//  * training: one wavefront per rating; an embedding row of d = 64*VPL doubles is VPL doubles
//    per lane (element i*64+lane, 512 contiguous bytes per wave instruction); the context rows of
//    an occurrence's centre + negatives stay in registers across all its window contexts;
//    rows race between wavefronts Hogwild-style with agent-scope loads and fp64 atomic adds.
//  * everything is fp64 like the reference's numpy arithmetic; gather/scatter bound — no MFMA.
// -ffp-contract=off (Makefile): every multiply and add rounds separately, as numpy does.
#include <cmath>

#include "n2v_bine.h"
#include "n2v_common.h"

namespace {

constexpr double kLn10 = 2.302585092994046;  // math.log(10, math.e), src/bine_train.py:300
constexpr int kMaxTrials = 1 << 16;          // proposal cap of a walk step (exit condition)
constexpr int kPoolTrials = 16;

__device__ __forceinline__ void philox4(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                        uint32_t (&out)[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

// xor-butterfly exchange of a double with lane ^ M: DPP quad permutes for M = 1, 2, ds_swizzle for 4, 8, 16
// (no address register, unlike ds_bpermute), a full shuffle for 32.
template <int M>
__device__ __forceinline__ double xchg(double x) {
    const uint64_t b = __builtin_bit_cast(uint64_t, x);
    int lo = (int)(uint32_t)b, hi = (int)(uint32_t)(b >> 32);
    if constexpr (M == 1) {
        lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true);
    } else if constexpr (M == 2) {
        lo = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xf, 0xf, true);
        hi = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xf, 0xf, true);
    } else if constexpr (M == 32) {
        lo = __shfl_xor(lo, 32);
        hi = __shfl_xor(hi, 32);
    } else {
        lo = __builtin_amdgcn_ds_swizzle(lo, (M << 10) | 0x1f);
        hi = __builtin_amdgcn_ds_swizzle(hi, (M << 10) | 0x1f);
    }
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
// Wave-wide sums of N values at once (the N butterflies are independent, so their latencies overlap);
// every lane ends with the same totals (a+b == b+a at every stage).
template <int N>
__device__ __forceinline__ void wave_sum_n(double (&x)[N]) {
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] += xchg<32>(x[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] += xchg<16>(x[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] += xchg<8>(x[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] += xchg<4>(x[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] += xchg<2>(x[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) x[j] += xchg<1>(x[j]);
}
__device__ __forceinline__ double wave_sum_d(double x) {
    double v[1] = {x};
    wave_sum_n<1>(v);
    return v[0];
}

// Values that are equal in all 64 lanes but reach us through vector registers (shuffles, vector loads):
// v_readfirstlane makes them scalar, so addresses derived from them live in SGPRs and loads from them
// become scalar loads — the training kernel's row data then has the VGPRs to itself.
__device__ __forceinline__ int32_t uni(int32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)(uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | lo);
}

// ------------------------------------------------------------------------------------ HITS
__global__ void __launch_bounds__(256) spmv_kernel(int64_t n_rows, const int64_t* __restrict__ row_ptr,
                                                   const int32_t* __restrict__ col, const double* __restrict__ w,
                                                   const double* __restrict__ x, double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int64_t b = row_ptr[r], e = row_ptr[r + 1];
    double s = 0.0;
    for (int64_t k = b + lane; k < e; k += 64) s += w[k] * x[col[k]];
    s = wave_sum_d(s);
    if (lane == 0) y[r] = s;
}

__device__ __forceinline__ double block_reduce(double v, bool is_max, double* sm) {
    const int t = threadIdx.x;
    sm[t] = v;
    __syncthreads();
    for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
        if (t < s) sm[t] = is_max ? fmax(sm[t], sm[t + s]) : sm[t] + sm[t + s];
        __syncthreads();
    }
    const double r = sm[0];
    __syncthreads();
    return r;
}

__global__ void __launch_bounds__(1024) hits_normalise_kernel(int64_t n, double* __restrict__ h, double* __restrict__ a,
                                                              const double* __restrict__ h_last, double* state) {
    __shared__ double sm[1024];
    double mh = 0.0, ma = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        mh = fmax(mh, h[i]);
        ma = fmax(ma, a[i]);
    }
    mh = block_reduce(mh, true, sm);
    ma = block_reduce(ma, true, sm);
    const double sh = 1.0 / mh, sa = 1.0 / ma;
    double err = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const double hv = h[i] * sh;
        h[i] = hv;
        a[i] = a[i] * sa;
        err += fabs(hv - h_last[i]);
    }
    err = block_reduce(err, false, sm);
    if (threadIdx.x == 0) state[0] = err;
}

__global__ void __launch_bounds__(1024) walk_counts_kernel(const double* __restrict__ a, int64_t lo, int64_t hi,
                                                           int32_t maxT, int32_t minT, int32_t* __restrict__ counts,
                                                           double* __restrict__ auth_out) {
    __shared__ double sm[1024];
    double mx = 0.0, mn = 100000.0;  // the reference's start values, src/bine_graph_utils.py:62
    for (int64_t i = lo + threadIdx.x; i < hi; i += 1024) {
        mx = fmax(mx, a[i]);
        mn = fmin(mn, a[i]);
    }
    mx = block_reduce(mx, true, sm);
    mn = -block_reduce(-mn, true, sm);
    const double span = mx - mn;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 1024) {
        const double s = span != 0.0 ? (a[i] - mn) / span : 0.0;
        if (auth_out) auth_out[i] = s;
        const int32_t c = (int32_t)ceil((double)maxT * s);
        counts[i] = c > minT ? c : minT;
    }
}

// ------------------------------------------------------------------------------------ walks
__global__ void __launch_bounds__(256) walk_len_kernel(const int64_t* __restrict__ row_ptr,
                                                       const int64_t* __restrict__ cum2,
                                                       const int32_t* __restrict__ walk_node, int64_t n_walks,
                                                       int64_t gw_base, double percentage, int32_t max_len,
                                                       uint64_t seed, int32_t* __restrict__ lens) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_walks) return;
    const uint64_t gw = (uint64_t)(gw_base + i);
    const int32_t node = walk_node[i];
    const int64_t rb = row_ptr[node], re = row_ptr[node + 1];
    const int64_t paths = cum2[re] - cum2[rb];
    int32_t len = 1;
    if (paths - (re - rb) > 0) {  // some two-hop path leaves the vertex (each middle vertex offers one path back)
        for (uint32_t t = 0; len < max_len; ++t) {
            uint32_t r[4];
            philox4(seed, (uint32_t)gw, (uint32_t)(gw >> 32), t, 0u, r);
            if (u53(r[0], r[1]) > percentage) ++len;  // `while ... random.random() > percentage`
            else break;
        }
    }
    lens[i] = len;
}

// Do the sorted ranges col[a_lo .. a_lo+a_n) and col[b_lo .. b_lo+b_n) share a value?  Answered by the
// whole wave: 64 elements of the shorter range are binary-searched in the longer one per round,
// with an early exit at the first hit.
__device__ __forceinline__ bool wave_any_common(const int32_t* __restrict__ col, int64_t a_lo, int32_t a_n, int64_t b_lo,
                                                int32_t b_n, int lane) {
    if (a_n > b_n) {
        const int64_t tl = a_lo; a_lo = b_lo; b_lo = tl;
        const int32_t tn = a_n; a_n = b_n; b_n = tn;
    }
    for (int32_t i0 = 0; i0 < a_n; i0 += 64) {
        const int32_t i = i0 + lane;
        bool found = false;
        if (i < a_n) {
            const int32_t x = col[a_lo + i];
            int32_t lo = 0, hi = b_n;
            while (lo < hi) {
                const int32_t mid = (lo + hi) >> 1;
                if (col[b_lo + mid] < x) lo = mid + 1;
                else hi = mid;
            }
            found = lo < b_n && col[b_lo + lo] == x;
        }
        if (__ballot(found) != 0ull) return true;
    }
    return false;
}

__global__ void __launch_bounds__(256) bine_walk_kernel(const int64_t* __restrict__ row_ptr,
                                                        const int32_t* __restrict__ col,
                                                        const int64_t* __restrict__ cum2,
                                                        const int32_t* __restrict__ walk_node,
                                                        const int64_t* __restrict__ walk_off, int64_t n_walks,
                                                        int64_t gw_base, uint64_t seed, int32_t* __restrict__ tokens) {
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t i = (int64_t)blockIdx.x * 4 + wv; i < n_walks; i += stride) {
        const uint64_t gw = (uint64_t)(gw_base + i);
        const int64_t off = walk_off[i];
        const int32_t len = (int32_t)(walk_off[i + 1] - off);
        int32_t cur = walk_node[i];
        if (lane == 0) tokens[off] = cur;
        for (int32_t t = 0; t + 1 < len; ++t) {
            const int64_t rb = row_ptr[cur], re = row_ptr[cur + 1];
            const int64_t base = cum2[rb];
            const int64_t paths = cum2[re] - base;
            int32_t next = cur;
            for (int trial = 0; trial < kMaxTrials; ++trial) {
                uint32_t r[4];
                philox4(seed, (uint32_t)gw, (uint32_t)(gw >> 32), (uint32_t)t, 1u + (uint32_t)trial, r);
                int64_t pick = (int64_t)floor(u53(r[0], r[1]) * (double)paths);
                if (pick >= paths) pick = paths - 1;
                int64_t lo = rb, hi = re - 1;  // the CSR entry whose path range holds `pick`
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (cum2[mid + 1] - base > pick) hi = mid;
                    else lo = mid + 1;
                }
                const int32_t mid_v = col[lo];
                const int32_t w = col[row_ptr[mid_v] + (pick - (cum2[lo] - base))];
                if (w == cur) continue;  // `while add_node == cur` (src/bine_graph.py:301)
                next = w;
                // keep the path only if mid_v is the FIRST common neighbour of cur and w: exactly one of the
                // |N(cur) & N(w)| paths reaching w survives, so every distinct w is equally likely
                const int64_t wb = row_ptr[w];
                int32_t plo = 0, phi = (int32_t)(row_ptr[w + 1] - wb);  // entries of row(w) below mid_v
                while (plo < phi) {
                    const int32_t pm = (plo + phi) >> 1;
                    if (col[wb + pm] < mid_v) plo = pm + 1;
                    else phi = pm;
                }
                if (!wave_any_common(col, rb, (int32_t)(lo - rb), wb, plo, lane)) break;
            }
            cur = next;
            if (lane == 0) tokens[off + t + 1] = cur;
        }
    }
}

// ------------------------------------------------------------------------------------ negative pools
__device__ __forceinline__ int lane_intersect(const int32_t* __restrict__ col, int64_t a_lo, int32_t a_n, int64_t b_lo,
                                              int32_t b_n) {
    if (a_n > b_n) {
        const int64_t tl = a_lo; a_lo = b_lo; b_lo = tl;
        const int32_t tn = a_n; a_n = b_n; b_n = tn;
    }
    int cnt = 0;
    for (int32_t i = 0; i < a_n; ++i) {
        const int32_t x = col[a_lo + i];
        int32_t lo = 0, hi = b_n;
        while (lo < hi) {
            const int32_t mid = (lo + hi) >> 1;
            if (col[b_lo + mid] < x) lo = mid + 1;
            else hi = mid;
        }
        cnt += (lo < b_n && col[b_lo + lo] == x) ? 1 : 0;
    }
    return cnt;
}

__global__ void __launch_bounds__(256) neg_pool_kernel(const int64_t* __restrict__ row_ptr,
                                                       const int32_t* __restrict__ col, int64_t side_lo, int64_t side_hi,
                                                       int64_t v_begin, int64_t v_end, int32_t pool_size,
                                                       double max_jaccard, uint64_t seed, int32_t* __restrict__ pool) {
    const int lane = threadIdx.x & 63;
    const int64_t v = v_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= v_end) return;
    const int64_t vb = row_ptr[v];
    const int32_t vn = (int32_t)(row_ptr[v + 1] - vb);
    const double side_n = (double)(side_hi - side_lo);
    for (int32_t s = lane; s < pool_size; s += 64) {
        int64_t c = v;
        for (int trial = 0; trial <= kPoolTrials; ++trial) {
            uint32_t r[4];
            philox4(seed, (uint32_t)v, (uint32_t)s, (uint32_t)trial, 0u, r);
            int64_t k = (int64_t)floor(u53(r[0], r[1]) * side_n);
            if (k >= side_hi - side_lo) k = side_hi - side_lo - 1;
            c = side_lo + k;
            if (c == v) continue;
            if (trial == kPoolTrials) break;  // give up thinning: keep this candidate
            const int64_t cb = row_ptr[c];
            const int32_t cn = (int32_t)(row_ptr[c + 1] - cb);
            const int mult = lane_intersect(col, vb, vn, cb, cn);
            if (!((double)mult > max_jaccard * (double)(vn + cn - mult))) break;
        }
        if (c == v) c = (v + 1 < side_hi) ? v + 1 : side_lo;  // all draws hit v itself: next vertex of the side
        pool[(v - v_begin) * pool_size + s] = (int32_t)c;
    }
}

// ------------------------------------------------------------------------------------ init
__global__ void __launch_bounds__(256) bine_init_kernel(double* __restrict__ emb, double* __restrict__ ctx, int64_t n,
                                                        int32_t dim, int32_t row_stride, uint64_t seed) {
    const int lane = threadIdx.x & 63;
    const int64_t job = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // row * 2 + table
    if (job >= 2 * n) return;
    const int64_t row = job >> 1;
    const uint32_t table = (uint32_t)(job & 1);
    double* out = (table ? ctx : emb) + row * row_stride;
    double ss = 0.0;
    for (int32_t c = lane; c < row_stride; c += 64) {
        double x = 0.0;
        if (c < dim) {
            uint32_t r[4];
            philox4(seed, (uint32_t)row, (uint32_t)((uint64_t)row >> 32), (uint32_t)(c >> 1), table, r);
            x = (c & 1) ? u53(r[2], r[3]) : u53(r[0], r[1]);
        }
        out[c] = x;
        ss += x * x;
    }
    ss = wave_sum_d(ss);
    const double norm = sqrt(ss);
    for (int32_t c = lane; c < dim; c += 64) out[c] = out[c] / norm;  // sklearn normalize(norm='l2')
}

// ------------------------------------------------------------------------------------ training
struct TrainArgs {
    const int32_t* edge_u;
    const int32_t* edge_v;
    const double* edge_w;
    const uint8_t* first;
    int64_t e_begin, e_end;
    double* emb;
    double* ctx;
    int32_t row_stride;
    const int64_t* occ_ptr;
    const int64_t* occ_pos;
    const int32_t* tokens;
    const int32_t* tok_walk;
    const int64_t* walk_off;
    const int32_t* pool;
    int32_t pool_size, ws, ns;
    double alpha, beta, gamma;
    double* state;
    int32_t iteration;
    uint64_t seed_occ, seed_neg;
};

template <int VPL>
struct DRow {
    double v[VPL];
};

template <int VPL, int MODE>
__device__ __forceinline__ DRow<VPL> load_row(const double* base, int64_t row, int stride, int lane) {
    DRow<VPL> r;
    const double* p = base + row * stride + lane;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        if constexpr (MODE == N2V_BINE_SEQUENTIAL) {
            r.v[i] = p[i * 64];
        } else {
            const uint64_t b = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p + i * 64), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
            r.v[i] = __builtin_bit_cast(double, b);
        }
    }
    return r;
}
// SEQUENTIAL: row = now (plain store).  PARALLEL: row += delta (fp64 atomic add at the memory side).
template <int VPL, int MODE>
__device__ __forceinline__ void commit_row(double* base, int64_t row, int stride, int lane, const DRow<VPL>& now,
                                           const DRow<VPL>& delta) {
    double* p = base + row * stride + lane;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        if constexpr (MODE == N2V_BINE_SEQUENTIAL) p[i * 64] = now.v[i];
        else __hip_atomic_fetch_add(p + i * 64, delta.v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <int VPL>
__device__ __forceinline__ double dot_row(const DRow<VPL>& a, const DRow<VPL>& b) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < VPL; ++i) s += a.v[i] * b.v[i];
    return wave_sum_d(s);
}

// Floyd's sample of m distinct values of [0, n): lane k ends up holding the k-th value.
// draw(k) is a 32-bit word; value = floor(word * (j+1) / 2^32) for j = n-m+k, replaced by j if taken.
template <typename F>
__device__ __forceinline__ uint32_t floyd_sample(uint32_t n, uint32_t m, int lane, F draw) {
    uint32_t mine = 0xFFFFFFFFu;
    for (uint32_t k = 0; k < m; ++k) {
        const uint32_t j = n - m + k;
        uint32_t t = __umulhi(draw(k), j + 1u);
        if (__ballot(lane < (int)k && mine == t) != 0ull) t = j;
        if (lane == (int)k) mine = t;
    }
    return mine;
}

// skip-gram block of vertex c (src/bine_train.py:462-474): sampled occurrences -> contexts, negatives ->
// skip_gram(c, z, negs) for every context z.  Returns nothing; adds to `loss` in the reference's order.
template <int VPL, int NT, int MODE>
__device__ __forceinline__ void node_block(const TrainArgs& a, int32_t c, double pa, double lam, int lane, double& loss,
                                           uint32_t& rows, uint32_t& rows_ref, double* th0) {
    const double pl = pa * lam;
    const int64_t ob = uni64(a.occ_ptr[c]);
    const uint32_t n_occ = (uint32_t)uni((int32_t)(a.occ_ptr[c + 1] - ob));
    const uint32_t m = n_occ < 10u ? n_occ : 10u;
    uint32_t ra[4], rb4[4], rc[4];
    philox4(a.seed_occ, (uint32_t)c, (uint32_t)a.iteration, 0u, 0u, ra);
    philox4(a.seed_occ, (uint32_t)c, (uint32_t)a.iteration, 1u, 0u, rb4);
    philox4(a.seed_occ, (uint32_t)c, (uint32_t)a.iteration, 2u, 0u, rc);
    const uint32_t my_occ = floyd_sample(n_occ, m, lane, [&](uint32_t k) {
        uint32_t w0 = ra[0], w1 = rb4[0], w2 = rc[0];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const bool hit = (k & 3u) == (uint32_t)q;
            w0 = hit ? ra[q] : w0;
            w1 = hit ? rb4[q] : w1;
            w2 = hit ? rc[q] : w2;
        }
        return k < 4u ? w0 : (k < 8u ? w1 : w2);
    });
    const int stride = a.row_stride;
    for (uint32_t k = 0; k < m; ++k) {
        const uint32_t idx = (uint32_t)uni(__shfl((int)my_occ, (int)k));
        const int64_t o = uni64(a.occ_pos[ob + idx]);
        const int32_t wk = uni(a.tok_walk[o]);
        const int64_t w0 = uni64(a.walk_off[wk]), w1 = uni64(a.walk_off[wk + 1]);
        const int64_t s = o - a.ws > w0 ? o - a.ws : w0;            // max(0, iter - win_size) within the walk
        const int64_t e = o + a.ws + 1 < w1 ? o + a.ws + 1 : w1;    // min(len, iter + win_size + 1)
        const int nwin = (int)(e - s);
        const int32_t mytok = lane < nwin ? a.tokens[s + lane] : -1;
        // negatives: distinct pool slots; dropped when in the window or already taken
        uint32_t rn0[4], rn1[4];
        philox4(a.seed_neg, (uint32_t)o, (uint32_t)((uint64_t)o >> 32), 0u, 0u, rn0);
        philox4(a.seed_neg, (uint32_t)o, (uint32_t)((uint64_t)o >> 32), 1u, 0u, rn1);
        const uint32_t m2 = (uint32_t)a.ns < (uint32_t)a.pool_size ? (uint32_t)a.ns : (uint32_t)a.pool_size;
        const uint32_t my_slot = floyd_sample((uint32_t)a.pool_size, m2, lane, [&](uint32_t q) {
            uint32_t w0 = rn0[0], w1 = rn1[0];
#pragma unroll
            for (int z = 1; z < 4; ++z) {
                w0 = ((q & 3u) == (uint32_t)z) ? rn0[z] : w0;
                w1 = ((q & 3u) == (uint32_t)z) ? rn1[z] : w1;
            }
            return q < 4u ? w0 : w1;
        });
        int32_t my_neg = -1;
        if (lane < (int)m2) my_neg = a.pool[(int64_t)c * a.pool_size + my_slot];
        int32_t tgt[NT];
#pragma unroll
        for (int z = 0; z < NT; ++z) tgt[z] = -1;
        int nt = 1;
        tgt[0] = c;
#pragma unroll
        for (int q = 0; q < NT - 1; ++q) {
            if (q < (int)m2) {
                const int32_t cand = uni(__shfl(my_neg, q));
                bool drop = cand < 0 || __ballot(mytok == cand) != 0ull;  // empty slot; in the window (src/bine_graph_utils.py:179-180)
#pragma unroll
                for (int z = 0; z < NT; ++z) drop = drop || (z < nt && tgt[z] == cand);  // I_z is a dict: one entry per vertex
                if (!drop) {
#pragma unroll
                    for (int z = 1; z < NT; ++z) if (z == nt) tgt[z] = cand;
                    ++nt;
                }
            }
        }
        // target rows live in registers across the window; in PARALLEL mode the values as loaded are parked in
        // this lane's own LDS slots so that the commit can add exactly (final - loaded) atomically
        DRow<VPL> th[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (j < nt) {
                th[j] = load_row<VPL, MODE>(a.ctx, tgt[j], stride, lane);
                if constexpr (MODE == N2V_BINE_PARALLEL) {
#pragma unroll
                    for (int i = 0; i < VPL; ++i) th0[(j * VPL + i) * 64 + lane] = th[j].v[i];
                }
            }
        }
        rows += 2u * (uint32_t)nt;
        // contexts: window positions whose token differs from the centre (`if walk[index] == walk[iter]: continue`)
        uint64_t todo = __ballot(lane < nwin && mytok != c);
        DRow<VPL> Vn;
        int32_t zn = -1;
        if (todo) {
            zn = uni(__shfl(mytok, (int)__builtin_ctzll(todo)));
            Vn = load_row<VPL, MODE>(a.emb, zn, stride, lane);
        }
        while (todo) {
            todo &= todo - 1;
            const int32_t z = zn;
            const DRow<VPL> V = Vn;
            bool fetched = false;
            if (todo) {
                zn = uni(__shfl(mytok, (int)__builtin_ctzll(todo)));
                // PARALLEL: fetch the next context row now, so its latency hides behind this context's arithmetic
                // and atomics (not when it is the same vertex: that row is about to change)
                if (MODE != N2V_BINE_SEQUENTIAL && zn != z) {
                    Vn = load_row<VPL, MODE>(a.emb, zn, stride, lane);
                    fetched = true;
                }
            }
            DRow<VPL> upd;
#pragma unroll
            for (int i = 0; i < VPL; ++i) upd.v[i] = 0.0;
            double l = 0.0;
            double dots[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                double sacc = 0.0;
#pragma unroll
                for (int i = 0; i < VPL; ++i) sacc += V.v[i] * th[j].v[i];
                dots[j] = j < nt ? sacc : 0.0;
            }
            wave_sum_n<NT>(dots);  // the targets' rows are distinct, so the dots do not depend on each other's updates
            // the sigmoid / log chain of target j is evaluated by lane j alone (one pass of exp and log for all
            // targets instead of one per target), then the coefficients and loss terms return by readlane
            double mydot = dots[0];
#pragma unroll
            for (int j = 1; j < NT; ++j) mydot = lane == j ? dots[j] : mydot;
            const double ind = lane == 0 ? 1.0 : 0.0;  // I_z: 1 for the centre, 0 for a negative
            const double X = fmax(mydot, 0.0);
            const double sig = 1.0 / (1.0 + exp(-X * 1.0));
            const double mycoef = pl * (ind - sig);
            const double one_m = 1.0 - sig;
            // math.log(0) raises -> the term is skipped (src/bine_train.py:268-271)
            const double myterm = one_m > 0.0 ? pa * (ind * log(sig) + (1.0 - ind) * log(one_m)) : 0.0;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (j < nt) {
                    const double coef = __builtin_bit_cast(double, uni64(__builtin_bit_cast(int64_t, __shfl(mycoef, j))));
#pragma unroll
                    for (int i = 0; i < VPL; ++i) {
                        upd.v[i] = upd.v[i] + coef * th[j].v[i];
                        th[j].v[i] = th[j].v[i] + coef * V.v[i];
                    }
                    l += __shfl(myterm, j);
                }
            }
            DRow<VPL> nowV;
#pragma unroll
            for (int i = 0; i < VPL; ++i) nowV.v[i] = V.v[i] + upd.v[i];
            commit_row<VPL, MODE>(a.emb, z, stride, lane, nowV, upd);
            loss += l;
            rows += 2u;
            rows_ref += 2u + 2u * (uint32_t)nt;  // skip_gram as written: V and every target row read and written per call
            if (todo && !fetched) Vn = load_row<VPL, MODE>(a.emb, zn, stride, lane);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
            if (j < nt) {
                if constexpr (MODE == N2V_BINE_PARALLEL_STORE) {
                    // context rows are written back whole at agent scope: a row is shared only when a vertex is,
                    // at that moment, also somebody's negative — the racing update is then lost, Hogwild-style
                    double* q = a.ctx + (int64_t)tgt[j] * stride + lane;
#pragma unroll
                    for (int i = 0; i < VPL; ++i)
                        __hip_atomic_store(reinterpret_cast<uint64_t*>(q + i * 64), __builtin_bit_cast(uint64_t, th[j].v[i]),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    DRow<VPL> dl;
#pragma unroll
                    for (int i = 0; i < VPL; ++i)
                        dl.v[i] = MODE == N2V_BINE_PARALLEL ? th[j].v[i] - th0[(j * VPL + i) * 64 + lane] : 0.0;
                    commit_row<VPL, MODE>(a.ctx, tgt[j], stride, lane, th[j], dl);
                }
            }
    }
}

template <int VPL, int NT, int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT > 5 || VPL > 4 ? 2 : (VPL == 4 ? 3 : 4))))
bine_train_kernel(TrainArgs a) {
    extern __shared__ double th0_all[];  // PARALLEL: [waves per block][NT][VPL][64]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* th0 = th0_all + wv * (NT * VPL * 64);
    const double lam = a.state[0];
    const double gl = a.gamma * lam;
    const int stride = a.row_stride;
    double loss = 0.0;
    uint32_t rows = 0;      // embedding rows this wave read + wrote (state[4])
    uint32_t rows_ref = 0;  // rows the reference's access pattern would move for the same work (state[5])
    // Work is handed out in chunks of kChunk consecutive ratings through a counter in state[6]: the cost of a
    // rating varies by orders of magnitude (first-seen vertices carry their skip-gram block, and those cluster
    // at the head of the list and at regular strides), so any static assignment leaves most waves idle.
    // One wave (SEQUENTIAL) simply takes the chunks in order.
    constexpr int64_t kChunk = 16;
    unsigned long long* next_chunk = reinterpret_cast<unsigned long long*>(a.state + 6);
    for (;;) {
        unsigned long long got = 0;
        if (lane == 0) got = __hip_atomic_fetch_add(next_chunk, (unsigned long long)kChunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t c0 = a.e_begin + uni64((int64_t)got);
        if (c0 >= a.e_end) break;
        const int64_t c1 = c0 + kChunk < a.e_end ? c0 + kChunk : a.e_end;
        // Consecutive ratings usually share their user (rating files are grouped by user): its embedding row is
        // carried in registers across them — the reference's own sequence U += g*V, U += g'*V', ... — and
        // committed once when the user changes or the chunk ends.
        int32_t held_u = -1;
        DRow<VPL> U, dU;
        for (int64_t e = c0; e < c1; ++e) {
            const int32_t u = uni(a.edge_u[e]), v = uni(a.edge_v[e]);
            const double w = __builtin_bit_cast(double, uni64(__builtin_bit_cast(int64_t, a.edge_w[e])));
            const int32_t f = uni((int32_t)a.first[e]);
            if (f != 0 && held_u >= 0) {  // a skip-gram block may touch any embedding row: hand ours back first
                commit_row<VPL, MODE>(a.emb, held_u, stride, lane, U, dU);
                held_u = -1;
            }
            for (int side = 0; side < 2; ++side)  // the user's block, then the item's (src/bine_train.py:462,475)
                if (f & (1 << side)) node_block<VPL, NT, MODE>(a, side ? v : u, side ? a.beta : a.alpha, lam, lane, loss, rows, rows_ref, th0);
            // KL_divergence (src/bine_train.py:277-309)
            if (u != held_u) {
                if (held_u >= 0) commit_row<VPL, MODE>(a.emb, held_u, stride, lane, U, dU);
                U = load_row<VPL, MODE>(a.emb, u, stride, lane);
#pragma unroll
                for (int i = 0; i < VPL; ++i) dU.v[i] = 0.0;
                held_u = u;
                rows += 2u;
            }
            const DRow<VPL> V = load_row<VPL, MODE>(a.emb, v, stride, lane);
            const double X = fmax(dot_row<VPL>(U, V), 0.0);
            const double sig = 1.0 / (1.0 + exp(-X * 1.0));
            const double g = gl * ((w * (1.0 - sig)) * 1.0 / kLn10);
            DRow<VPL> dv, nv;
#pragma unroll
            for (int i = 0; i < VPL; ++i) {
                const double du = g * V.v[i];
                dv.v[i] = g * U.v[i];
                nv.v[i] = V.v[i] + dv.v[i];
                U.v[i] = U.v[i] + du;
                dU.v[i] = dU.v[i] + du;
            }
            commit_row<VPL, MODE>(a.emb, v, stride, lane, nv, dv);
            loss += a.gamma * w * log(sig);
            rows += 2u;
            rows_ref += 4u;
        }
        if (held_u >= 0) commit_row<VPL, MODE>(a.emb, held_u, stride, lane, U, dU);
    }
    if (lane == 0) {
        __hip_atomic_fetch_add(&a.state[1], loss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.state[4], (double)rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&a.state[5], (double)rows_ref, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void lambda_step_kernel(double* state, double epsilon) {
    const double lam = state[0], loss = state[1], last = state[2];
    state[0] = last > loss ? lam * 1.05 : lam * 0.95;
    state[3] = fabs(loss - last) < epsilon ? 1.0 : 0.0;
    state[2] = loss;
    state[1] = 0.0;
    *reinterpret_cast<unsigned long long*>(state + 6) = 0ull;  // work counter of the next pass
}

template <int VPL, int NT>
int launch_train(const TrainArgs& a, int mode, int max_blocks, hipStream_t st) {
    if (mode == N2V_BINE_SEQUENTIAL) {
        hipLaunchKernelGGL((bine_train_kernel<VPL, NT, N2V_BINE_SEQUENTIAL>), dim3(1), dim3(64), 0, st, a);
    } else {
        const int64_t n = a.e_end - a.e_begin;
        int64_t blocks = (n + 3) / 4;
        const int64_t cap = max_blocks > 0 ? max_blocks : 2048;
        if (blocks > cap) blocks = cap;
        if (mode == N2V_BINE_PARALLEL)
            hipLaunchKernelGGL((bine_train_kernel<VPL, NT, N2V_BINE_PARALLEL>), dim3((unsigned)blocks), dim3(256),
                               4 * NT * VPL * 64 * sizeof(double), st, a);
        else
            hipLaunchKernelGGL((bine_train_kernel<VPL, NT, N2V_BINE_PARALLEL_STORE>), dim3((unsigned)blocks), dim3(256), 0,
                               st, a);
    }
    return n2v::check_launch("n2v_bine_train_pass");
}

}  // namespace

// ==================================================================================== C-ABI
extern "C" int n2v_bine_spmv(int64_t n_rows, const int64_t* row_ptr, const int32_t* col, const double* w,
                             const double* x, double* y, void* stream) {
    if (n_rows < 0 || !row_ptr || !x || !y) return n2v::fail(N2V_ERR_INVALID, "n2v_bine_spmv: bad argument");
    if (n_rows == 0) return N2V_OK;
    if (!col || !w) return n2v::fail(N2V_ERR_INVALID, "n2v_bine_spmv: null col/w");
    hipLaunchKernelGGL(spmv_kernel, dim3(n2v::grid_for(n_rows, 4)), dim3(256), 0, (hipStream_t)stream, n_rows, row_ptr,
                       col, w, x, y);
    return n2v::check_launch("n2v_bine_spmv");
}

extern "C" int n2v_bine_hits_normalise(int64_t n, double* h, double* a, const double* h_last, double* state,
                                       void* stream) {
    if (n <= 0 || !h || !a || !h_last || !state) return n2v::fail(N2V_ERR_INVALID, "n2v_bine_hits_normalise: bad argument");
    hipLaunchKernelGGL(hits_normalise_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, h, a, h_last, state);
    return n2v::check_launch("n2v_bine_hits_normalise");
}

extern "C" int n2v_bine_walk_counts(const double* a, int64_t lo, int64_t hi, int32_t maxT, int32_t minT,
                                    int32_t* counts, double* auth_out, void* stream) {
    if (!a || !counts || lo < 0 || hi < lo || maxT < 0 || minT < 0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_bine_walk_counts: bad argument");
    if (hi == lo) return N2V_OK;
    hipLaunchKernelGGL(walk_counts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, a, lo, hi, maxT, minT, counts,
                       auth_out);
    return n2v::check_launch("n2v_bine_walk_counts");
}

extern "C" int n2v_bine_walk_lengths(const int64_t* row_ptr, const int64_t* cum2, const int32_t* walk_node,
                                     int64_t n_walks, int64_t gw_base, double percentage, int32_t max_len,
                                     uint64_t seed, int32_t* lens, void* stream) {
    if (n_walks < 0 || gw_base < 0 || max_len < 1 || !(percentage >= 0.0))
        return n2v::fail(N2V_ERR_INVALID, "n2v_bine_walk_lengths: bad size");
    if (n_walks == 0) return N2V_OK;
    if (!row_ptr || !cum2 || !walk_node || !lens) return n2v::fail(N2V_ERR_INVALID, "n2v_bine_walk_lengths: null pointer");
    hipLaunchKernelGGL(walk_len_kernel, dim3(n2v::grid_for(n_walks, 256)), dim3(256), 0, (hipStream_t)stream, row_ptr,
                       cum2, walk_node, n_walks, gw_base, percentage, max_len, seed, lens);
    return n2v::check_launch("n2v_bine_walk_lengths");
}

extern "C" int n2v_bine_walk(const int64_t* row_ptr, const int32_t* col, const int64_t* cum2,
                             const int32_t* walk_node, const int64_t* walk_off, int64_t n_walks, int64_t gw_base,
                             uint64_t seed, int32_t* tokens, void* stream) {
    if (n_walks < 0 || gw_base < 0) return n2v::fail(N2V_ERR_INVALID, "n2v_bine_walk: bad size");
    if (n_walks == 0) return N2V_OK;
    if (!row_ptr || !col || !cum2 || !walk_node || !walk_off || !tokens)
        return n2v::fail(N2V_ERR_INVALID, "n2v_bine_walk: null pointer");
    int64_t blocks = (n_walks + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(bine_walk_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, row_ptr, col, cum2,
                       walk_node, walk_off, n_walks, gw_base, seed, tokens);
    return n2v::check_launch("n2v_bine_walk");
}

extern "C" int n2v_bine_neg_pools(const int64_t* row_ptr, const int32_t* col, int64_t side_lo, int64_t side_hi,
                                  int64_t v_begin, int64_t v_end, int32_t pool_size, double max_jaccard,
                                  uint64_t seed, int32_t* pool, void* stream) {
    if (side_lo < 0 || side_hi <= side_lo || v_begin < side_lo || v_end > side_hi || v_end < v_begin || pool_size < 1 ||
        side_hi >= ((int64_t)1 << 31))
        return n2v::fail(N2V_ERR_INVALID, "n2v_bine_neg_pools: bad range");
    if (v_end == v_begin) return N2V_OK;
    if (!row_ptr || !col || !pool) return n2v::fail(N2V_ERR_INVALID, "n2v_bine_neg_pools: null pointer");
    hipLaunchKernelGGL(neg_pool_kernel, dim3(n2v::grid_for(v_end - v_begin, 4)), dim3(256), 0, (hipStream_t)stream,
                       row_ptr, col, side_lo, side_hi, v_begin, v_end, pool_size, max_jaccard, seed, pool);
    return n2v::check_launch("n2v_bine_neg_pools");
}

extern "C" int n2v_bine_init(double* emb, double* ctx, int64_t n, int32_t dim, int32_t row_stride, uint64_t seed,
                             void* stream) {
    if (!emb || !ctx || n < 0 || dim < 1 || row_stride < dim || (row_stride % 64) != 0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_bine_init: bad argument");
    if (n == 0) return N2V_OK;
    hipLaunchKernelGGL(bine_init_kernel, dim3(n2v::grid_for(2 * n, 4)), dim3(256), 0, (hipStream_t)stream, emb, ctx, n,
                       dim, row_stride, seed);
    return n2v::check_launch("n2v_bine_init");
}

extern "C" int n2v_bine_train_pass(const int32_t* edge_u, const int32_t* edge_v, const double* edge_w,
                                   const uint8_t* first, int64_t e_begin, int64_t e_end, double* emb, double* ctx,
                                   int32_t dim, int32_t row_stride, const int64_t* occ_ptr, const int64_t* occ_pos,
                                   const int32_t* tokens, const int32_t* tok_walk, const int64_t* walk_off,
                                   const int32_t* pool, int32_t pool_size, int32_t ws, int32_t ns, double alpha,
                                   double beta, double gamma, double* state, int32_t iteration, uint64_t seed_occ,
                                   uint64_t seed_neg, int32_t mode, int32_t max_blocks, void* stream) {
    if (e_begin < 0 || e_end < e_begin || dim < 1 || row_stride < dim || ws < 1 || ws > 31 || ns < 0 || ns > 7 ||
        pool_size < 1 || iteration < 0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_bine_train_pass: bad size (dim %d stride %d ws %d ns %d pool %d)", (int)dim,
                         (int)row_stride, (int)ws, (int)ns, (int)pool_size);
    if (row_stride != 64 && row_stride != 128 && row_stride != 256 && row_stride != 512)
        return n2v::fail(N2V_ERR_INVALID, "n2v_bine_train_pass: row_stride must be 64, 128, 256 or 512 (got %d)", (int)row_stride);
    if (mode != N2V_BINE_SEQUENTIAL && mode != N2V_BINE_PARALLEL && mode != N2V_BINE_PARALLEL_STORE)
        return n2v::fail(N2V_ERR_INVALID, "n2v_bine_train_pass: unknown mode %d", (int)mode);
    if (e_end == e_begin) return N2V_OK;
    if (!edge_u || !edge_v || !edge_w || !first || !emb || !ctx || !occ_ptr || !occ_pos || !tokens || !tok_walk ||
        !walk_off || !pool || !state)
        return n2v::fail(N2V_ERR_INVALID, "n2v_bine_train_pass: null pointer");
    TrainArgs a{edge_u, edge_v, edge_w, first, e_begin, e_end, emb, ctx, row_stride, occ_ptr, occ_pos, tokens, tok_walk,
                walk_off, pool, pool_size, ws, ns, alpha, beta, gamma, state, iteration, seed_occ, seed_neg};
    hipStream_t st = (hipStream_t)stream;
    const bool small = ns <= 4;
    switch (row_stride) {
        case 64: return small ? launch_train<1, 5>(a, mode, max_blocks, st) : launch_train<1, 8>(a, mode, max_blocks, st);
        case 128: return small ? launch_train<2, 5>(a, mode, max_blocks, st) : launch_train<2, 8>(a, mode, max_blocks, st);
        case 256: return small ? launch_train<4, 5>(a, mode, max_blocks, st) : launch_train<4, 8>(a, mode, max_blocks, st);
        default: return small ? launch_train<8, 5>(a, mode, max_blocks, st) : launch_train<8, 8>(a, mode, max_blocks, st);
    }
}

extern "C" int n2v_bine_lambda_step(double* state, double epsilon, void* stream) {
    if (!state) return n2v::fail(N2V_ERR_INVALID, "n2v_bine_lambda_step: null state");
    hipLaunchKernelGGL(lambda_step_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, epsilon);
    return n2v::check_launch("n2v_bine_lambda_step");
}
