// MinHash LSH-forest negative pools of the BiNE path — gfx950 (MI355X) kernels and C-ABI (include/n2v_bine.h).
//
// Replaces src/bine_lsh.py:7-51 (construct_lsh / call_get_negs_by_lsh) and the two datasketch 1.2.5 classes it
// drives (MinHash, MinHashLSHForest; requirements.txt:11), restated from their published source (DESIGN.md 4.7;
// the tests hold an independent dictionary/sorted-list restatement of the same source to compare with).
//
// Mapping to the machine
//  * SHA-1 of every vertex label once (one lane per label), not once per (vertex, neighbour) as the reference's
//    `temp.update(d.encode('utf8'))` loop does: the 32-bit value hv only depends on the neighbour;
//  * signatures: one wavefront per vertex, two permutations per lane, the neighbours' hv broadcast — 128 uint32
//    per vertex written as two 256-B lines;
//  * the eight prefix trees are eight sorted orders of the side's vertices (the host sorts, torch plumbing); a
//    query never searches: the vertex is in the forest, so the keys sharing its first r values are the contiguous
//    range [lo_r, hi_r) around its own sorted position, nested in r — each level only scans the two new wings;
//  * the result set lives in a 512-slot LDS hash per wavefront; 64 candidates are tested per step, new ones are
//    ranked with ballot/popcount so that the k-th key stops the query exactly where the reference's loop returns;
//  * the greedy "visited" sweep of call_get_negs_by_lsh is resolved in rounds over the reverse lists (a vertex waits
//    only for earlier vertices that list it);
//  * pools: one wavefront per cluster, the excluded set is a bitmap in a per-workgroup slice of HBM (L2 resident),
//    candidates are drawn 64 at a time and taken in lane order.
// Integer work, request bound; no MFMA.
#include "n2v_bine.h"
#include "n2v_common.h"

namespace {

constexpr uint64_t kMersenne = (1ull << 61) - 1ull;
constexpr int kPerm = N2V_LSH_NUM_PERM;   // 128
constexpr int kTrees = N2V_LSH_TREES;     // 8
constexpr int kDepth = kPerm / kTrees;    // 16
constexpr int kSetSlots = 512;            // LDS hash per wavefront; k <= 256

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

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

// ------------------------------------------------------------------------------------ SHA-1 (FIPS 180-4)
__device__ __forceinline__ uint32_t rotl(uint32_t x, int s) { return (x << s) | (x >> (32 - s)); }

// byte p of the padded message of a label of `len` bytes (padded length `total`)
__device__ __forceinline__ uint32_t padded_byte(const uint8_t* __restrict__ s, int32_t len, int32_t total, int32_t p) {
    if (p < len) return s[p];
    if (p == len) return 0x80u;
    if (p >= total - 8) {
        const uint64_t bits = (uint64_t)len * 8ull;
        return (uint32_t)(bits >> (8 * (total - 1 - p))) & 0xFFu;
    }
    return 0u;
}

__global__ void __launch_bounds__(256) sha1_labels_kernel(const uint8_t* __restrict__ bytes, int32_t width,
                                                          const int32_t* __restrict__ lens, int64_t n,
                                                          uint32_t* __restrict__ hv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t* s = bytes + i * width;
    const int32_t len = lens[i];
    const int32_t total = ((len + 9 + 63) / 64) * 64;
    uint32_t h0 = 0x67452301u, h1 = 0xEFCDAB89u, h2 = 0x98BADCFEu, h3 = 0x10325476u, h4 = 0xC3D2E1F0u;
    for (int32_t blk = 0; blk < total; blk += 64) {
        uint32_t w[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int32_t p = blk + 4 * t;
            w[t] = (padded_byte(s, len, total, p) << 24) | (padded_byte(s, len, total, p + 1) << 16) |
                   (padded_byte(s, len, total, p + 2) << 8) | padded_byte(s, len, total, p + 3);
        }
        uint32_t a = h0, b = h1, c = h2, d = h3, e = h4;
#pragma unroll
        for (int t = 0; t < 80; ++t) {
            if (t >= 16) {
                const uint32_t x = w[(t - 3) & 15] ^ w[(t - 8) & 15] ^ w[(t - 14) & 15] ^ w[t & 15];
                w[t & 15] = rotl(x, 1);
            }
            uint32_t f, k;
            if (t < 20) { f = (b & c) | (~b & d); k = 0x5A827999u; }
            else if (t < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1u; }
            else if (t < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDCu; }
            else { f = b ^ c ^ d; k = 0xCA62C1D6u; }
            const uint32_t tmp = rotl(a, 5) + f + e + k + w[t & 15];
            e = d; d = c; c = rotl(b, 30); b = a; a = tmp;
        }
        h0 += a; h1 += b; h2 += c; h3 += d; h4 += e;
    }
    // struct.unpack('<I', digest[:4]): the digest's first four bytes are h0 big endian
    hv[i] = __builtin_bswap32(h0);
}

// ------------------------------------------------------------------------------------ signatures
// x mod (2^61 - 1) for any 64-bit x: fold the top three bits once, one conditional subtract
__device__ __forceinline__ uint64_t mod_mersenne(uint64_t x) {
    uint64_t y = (x & kMersenne) + (x >> 61);
    return y >= kMersenne ? y - kMersenne : y;
}

__global__ void __launch_bounds__(256) minhash_kernel(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                      const uint32_t* __restrict__ hv, const uint64_t* __restrict__ perm_a,
                                                      const uint64_t* __restrict__ perm_b, int64_t v_begin, int64_t v_end,
                                                      uint32_t* __restrict__ sig) {
    const int lane = threadIdx.x & 63;
    const int64_t v = v_begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= v_end) return;
    const uint64_t a0 = perm_a[lane], a1 = perm_a[64 + lane], b0 = perm_b[lane], b1 = perm_b[64 + lane];
    uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu;
    const int64_t e0 = row_ptr[v], e1 = row_ptr[v + 1];
    for (int64_t base = e0; base < e1; base += 64) {
        const int cnt = (int)((e1 - base) < 64 ? (e1 - base) : 64);
        const uint32_t mine = lane < cnt ? hv[col[base + lane]] : 0u;
        for (int j = 0; j < cnt; ++j) {
            const uint64_t h = (uint64_t)(uint32_t)__shfl((int)mine, j);
            const uint32_t p0 = (uint32_t)mod_mersenne(a0 * h + b0);   // & (2^32 - 1)
            const uint32_t p1 = (uint32_t)mod_mersenne(a1 * h + b1);
            m0 = p0 < m0 ? p0 : m0;
            m1 = p1 < m1 ? p1 : m1;
        }
    }
    uint32_t* out = sig + (v - v_begin) * kPerm;
    out[lane] = m0;
    out[64 + lane] = m1;
}

// ------------------------------------------------------------------------------------ forest query
struct SetScratch {
    int32_t slot[kSetSlots];
};

__device__ __forceinline__ uint32_t set_home(int32_t key) { return ((uint32_t)key * 2654435761u) >> (32 - 9); }

__device__ __forceinline__ bool set_contains(const SetScratch& s, int32_t key) {
    uint32_t h = set_home(key);
    for (;;) {
        const int32_t x = __hip_atomic_load(&s.slot[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        if (x == key) return true;
        if (x < 0) return false;
        h = (h + 1) & (kSetSlots - 1);
    }
}
__device__ __forceinline__ void set_insert(SetScratch& s, int32_t key) {
    uint32_t h = set_home(key);
    for (;;) {
        int32_t expect = -1;
        if (__hip_atomic_compare_exchange_strong(&s.slot[h], &expect, key, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WAVEFRONT))
            return;
        h = (h + 1) & (kSetSlots - 1);
    }
}

// order: int32[kTrees][n] sorted position -> vertex; lo/hi: int32[n][kTrees][kDepth], [.][t][r-1] = the range of sorted
// positions of tree t whose first r values equal the vertex's.  sim: int32[n][k] (new keys in forest order), sim_n.
__global__ void __launch_bounds__(256) forest_query_kernel(const int32_t* __restrict__ order, const int32_t* __restrict__ lo,
                                                           const int32_t* __restrict__ hi, int64_t n, int32_t k,
                                                           int32_t* __restrict__ sim, int32_t* __restrict__ sim_n) {
    __shared__ SetScratch sets[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t v = (int64_t)blockIdx.x * 4 + wv;
    if (v >= n) return;
    SetScratch& set = sets[wv];
    for (int i = lane; i < kSetSlots; i += 64) set.slot[i] = -1;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int32_t* out = sim + v * k;
    int cnt = 0;
    const int32_t* vlo = lo + v * (kTrees * kDepth);
    const int32_t* vhi = hi + v * (kTrees * kDepth);
    for (int r = kDepth; r >= 1 && cnt < k; --r) {
        for (int t = 0; t < kTrees && cnt < k; ++t) {
            const int32_t nlo = uni(vlo[t * kDepth + r - 1]), nhi = uni(vhi[t * kDepth + r - 1]);
            // wings that level r adds to level r + 1's range (the whole range at the deepest level)
            int32_t w_lo[2], w_hi[2];
            if (r == kDepth) {
                w_lo[0] = nlo; w_hi[0] = nhi; w_lo[1] = 0; w_hi[1] = 0;
            } else {
                w_lo[0] = nlo; w_hi[0] = uni(vlo[t * kDepth + r]);
                w_lo[1] = uni(vhi[t * kDepth + r]); w_hi[1] = nhi;
            }
            const int32_t* ord = order + (int64_t)t * n;
            for (int w = 0; w < 2 && cnt < k; ++w) {
                for (int32_t base = w_lo[w]; base < w_hi[w] && cnt < k; base += 64) {
                    const int32_t i = base + lane;
                    const int32_t key = i < w_hi[w] ? ord[i] : -1;
                    const bool fresh = key >= 0 && !set_contains(set, key);
                    const uint64_t mask = __ballot(fresh);
                    const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                    const int room = k - cnt;
                    if (fresh && rank < room) {
                        out[cnt + rank] = key;
                        set_insert(set, key);
                    }
                    const int got = __popcll(mask);
                    cnt += got < room ? got : room;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    }
    if (lane == 0) sim_n[v] = cnt;
}

// ------------------------------------------------------------------------------------ clusters ("visited" sweep)
// One round: vertex i (owner[i] < 0) walks its reverse list (the earlier vertices l < i that list i, ascending): an
// unresolved l stops the walk until a later round, a leader l (owner[l] == l) owns i, a follower is skipped; the
// end of the list makes i a leader.  Owners only ever change from -1 to their final value.
__global__ void __launch_bounds__(256) leader_round_kernel(const int64_t* __restrict__ rev_ptr, const int32_t* __restrict__ rev_src,
                                                           int64_t n, int32_t* owner, int32_t* __restrict__ unresolved) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (__hip_atomic_load(&owner[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) return;
    int32_t mine = (int32_t)i;
    for (int64_t e = rev_ptr[i]; e < rev_ptr[i + 1]; ++e) {
        const int32_t l = rev_src[e];
        const int32_t s = __hip_atomic_load(&owner[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s < 0) { mine = -1; break; }
        if (s == l) { mine = l; break; }
    }
    if (mine >= 0) __hip_atomic_store(&owner[i], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else atomicAdd(unresolved, 1);
}

// ------------------------------------------------------------------------------------ pools
__device__ __forceinline__ bool bit_test(const uint32_t* bm, int32_t c) {
    return (__hip_atomic_load(&bm[c >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (c & 31)) & 1u;
}
__device__ __forceinline__ void bit_set(uint32_t* bm, int32_t c) {
    __hip_atomic_fetch_or(&bm[c >> 5], 1u << (c & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void word_clear(uint32_t* bm, int32_t c) {
    __hip_atomic_store(&bm[c >> 5], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wavefront per workgroup, leaders lead[blockIdx.x], lead[blockIdx.x + gridDim.x], ...; bitmap: words uint32 per
// workgroup, all zero on entry and on exit.  pool row of a leader (row index = its local id): id_base + vertex, or -1.
__global__ void __launch_bounds__(64) lsh_pool_kernel(const int32_t* __restrict__ sim, const int32_t* __restrict__ sim_n,
                                                      int32_t k, const int32_t* __restrict__ lead, int64_t n_lead,
                                                      int32_t n_side, int32_t pool_size, uint64_t seed, int32_t id_base,
                                                      uint32_t* bitmap, int64_t words, int32_t* __restrict__ pool) {
    const int lane = threadIdx.x;
    uint32_t* bm = bitmap + (int64_t)blockIdx.x * words;
    for (int64_t q = blockIdx.x; q < n_lead; q += gridDim.x) {
        const int32_t l = uni(lead[q]);
        const int32_t sn = uni(sim_n[l]);
        const int32_t* sl = sim + (int64_t)l * k;
        // excluded = sim(l) | U_{j in sim(l)} sim(j)
        for (int32_t a = lane; a < sn; a += 64) bit_set(bm, sl[a]);
        for (int32_t a = 0; a < sn; ++a) {
            const int32_t j = uni(sl[a]);
            const int32_t jn = uni(sim_n[j]);
            const int32_t* sj = sim + (int64_t)j * k;
            for (int32_t b = lane; b < jn; b += 64) bit_set(bm, sj[b]);
        }
        __threadfence();
        __builtin_amdgcn_wave_barrier();
        int32_t* row = pool + (int64_t)l * pool_size;
        bool all_of_it = false;
        if ((int64_t)n_side <= (int64_t)k * (k + 1) + pool_size) {   // the complement may be short: count it
            int gone = 0;
            for (int64_t w = lane; w < words; w += 64)
                gone += __popc(__hip_atomic_load(&bm[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            for (int s = 32; s >= 1; s >>= 1) gone += __shfl_xor(gone, s);
            all_of_it = n_side - gone <= pool_size;
        }
        int cnt = 0;
        if (all_of_it) {   // every vertex that is left, ascending, then -1
            for (int32_t base = 0; base < n_side; base += 64) {
                const int32_t c = base + lane;
                const bool keep = c < n_side && !bit_test(bm, c);
                const uint64_t mask = __ballot(keep);
                if (keep) row[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = id_base + c;
                cnt += __popcll(mask);
            }
            for (int32_t s = cnt + lane; s < pool_size; s += 64) row[s] = -1;
        } else {
            for (uint32_t rnd = 0; cnt < pool_size; ++rnd) {
                uint32_t r[4];
                philox4(seed, (uint32_t)l, rnd, (uint32_t)lane, 0u, r);
                int32_t c = (int32_t)floor(u53(r[0], r[1]) * (double)n_side);
                c = c < n_side ? c : n_side - 1;
                bool ok = !bit_test(bm, c);
                for (int d = 0; d < 63; ++d) {   // a lower lane of this round proposes the same vertex
                    const int32_t other = __shfl(c, d);
                    ok = ok && !(d < lane && other == c);
                }
                const uint64_t mask = __ballot(ok);
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                const int room = pool_size - cnt;
                if (ok && rank < room) {
                    row[cnt + rank] = id_base + c;
                    bit_set(bm, c);
                }
                const int got = __popcll(mask);
                cnt += got < room ? got : room;
                __threadfence();
                __builtin_amdgcn_wave_barrier();
            }
            for (int32_t s = lane; s < pool_size; s += 64)   // the row as L2 holds it (other lanes wrote it)
                word_clear(bm, __hip_atomic_load(&row[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - id_base);
        }
        // leave the bitmap zero for the next leader
        for (int32_t a = lane; a < sn; a += 64) word_clear(bm, sl[a]);
        for (int32_t a = 0; a < sn; ++a) {
            const int32_t j = uni(sl[a]);
            const int32_t jn = uni(sim_n[j]);
            const int32_t* sj = sim + (int64_t)j * k;
            for (int32_t b = lane; b < jn; b += 64) word_clear(bm, sj[b]);
        }
        __threadfence();
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" int n2v_lsh_sha1_labels(const uint8_t* bytes, int32_t width, const int32_t* lens, int64_t n, uint32_t* hv,
                                   void* stream) {
    if (n < 0 || width < 1) return n2v::fail(N2V_ERR_INVALID, "n2v_lsh_sha1_labels: bad size (n %lld width %d)", (long long)n, (int)width);
    if (n == 0) return N2V_OK;
    if (!bytes || !lens || !hv) return n2v::fail(N2V_ERR_INVALID, "n2v_lsh_sha1_labels: null pointer");
    hipLaunchKernelGGL(sha1_labels_kernel, dim3(n2v::grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, bytes, width,
                       lens, n, hv);
    return n2v::check_launch("n2v_lsh_sha1_labels");
}

extern "C" int n2v_lsh_minhash(const int64_t* row_ptr, const int32_t* col, const uint32_t* hv, const uint64_t* perm_a,
                               const uint64_t* perm_b, int64_t v_begin, int64_t v_end, uint32_t* sig, void* stream) {
    if (v_begin < 0 || v_end < v_begin) return n2v::fail(N2V_ERR_INVALID, "n2v_lsh_minhash: bad range");
    if (v_end == v_begin) return N2V_OK;
    if (!row_ptr || !col || !hv || !perm_a || !perm_b || !sig) return n2v::fail(N2V_ERR_INVALID, "n2v_lsh_minhash: null pointer");
    hipLaunchKernelGGL(minhash_kernel, dim3(n2v::grid_for(v_end - v_begin, 4)), dim3(256), 0, (hipStream_t)stream, row_ptr,
                       col, hv, perm_a, perm_b, v_begin, v_end, sig);
    return n2v::check_launch("n2v_lsh_minhash");
}

extern "C" int n2v_lsh_forest_query(const int32_t* order, const int32_t* lo, const int32_t* hi, int64_t n, int32_t k,
                                    int32_t* sim, int32_t* sim_n, void* stream) {
    if (n < 0 || n > 0x7FFFFFFF || k < 1 || k > kSetSlots / 2)
        return n2v::fail(N2V_ERR_INVALID, "n2v_lsh_forest_query: bad size (n %lld, k %d; k <= %d)", (long long)n, (int)k, kSetSlots / 2);
    if (n == 0) return N2V_OK;
    if (!order || !lo || !hi || !sim || !sim_n) return n2v::fail(N2V_ERR_INVALID, "n2v_lsh_forest_query: null pointer");
    hipLaunchKernelGGL(forest_query_kernel, dim3(n2v::grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, order, lo, hi, n,
                       k, sim, sim_n);
    return n2v::check_launch("n2v_lsh_forest_query");
}

extern "C" int n2v_lsh_leader_round(const int64_t* rev_ptr, const int32_t* rev_src, int64_t n, int32_t* owner,
                                    int32_t* unresolved, void* stream) {
    if (n < 0 || n > 0x7FFFFFFF) return n2v::fail(N2V_ERR_INVALID, "n2v_lsh_leader_round: bad size");
    if (n == 0) return N2V_OK;
    if (!rev_ptr || !owner || !unresolved) return n2v::fail(N2V_ERR_INVALID, "n2v_lsh_leader_round: null pointer");
    hipLaunchKernelGGL(leader_round_kernel, dim3(n2v::grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, rev_ptr, rev_src,
                       n, owner, unresolved);
    return n2v::check_launch("n2v_lsh_leader_round");
}

extern "C" int n2v_lsh_pools(const int32_t* sim, const int32_t* sim_n, int32_t k, const int32_t* lead, int64_t n_lead,
                             int32_t n_side, int32_t pool_size, uint64_t seed, int32_t id_base, uint32_t* bitmap,
                             int64_t words_per_group, int32_t n_groups, int32_t* pool, void* stream) {
    if (n_lead < 0 || n_side < 1 || k < 1 || pool_size < 1 || n_groups < 1 || words_per_group < ((int64_t)n_side + 31) / 32)
        return n2v::fail(N2V_ERR_INVALID, "n2v_lsh_pools: bad size (n_side %d, k %d, pool %d, groups %d, words %lld)", (int)n_side,
                         (int)k, (int)pool_size, (int)n_groups, (long long)words_per_group);
    if (n_lead == 0) return N2V_OK;
    if (!sim || !sim_n || !lead || !bitmap || !pool) return n2v::fail(N2V_ERR_INVALID, "n2v_lsh_pools: null pointer");
    const int64_t grid = n_lead < n_groups ? n_lead : n_groups;
    hipLaunchKernelGGL(lsh_pool_kernel, dim3((unsigned)grid), dim3(64), 0, (hipStream_t)stream, sim, sim_n, k, lead, n_lead,
                       n_side, pool_size, seed, id_base, bitmap, words_per_group, pool);
    return n2v::check_launch("n2v_lsh_pools");
}
