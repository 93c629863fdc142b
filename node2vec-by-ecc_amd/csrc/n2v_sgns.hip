// Skip-gram with negative sampling over a device-resident walk corpus — gfx950 kernels.
//
// Replaces what `learn_embeddings` hands to gensim 3.2.0 (src/main.py:82-90:
// Word2Vec(walks, size=d, window=w, min_count=0, sg=1, iter=...), defaults negative=5,
// alpha .025 -> .0001, sample=1e-3).  gensim's source is not part of the reference tree;
// the update rule below restates its public `fast_sentence_sg_neg` / `train_batch_sg`
// (word2vec_inner.pyx) as summarised in SURVEY.md 8(a) row 9:
//
//   per sentence: drop sub-sampled words; per position i draw b in [0, window); for every
//   j in [i-window+b, i+window-b], j != i:   input row h = syn0[word_j]; targets = word_i
//   (label 1) + `negative` draws from the unigram^0.75 cum-table (a draw equal to word_i is
//   skipped); f = <h, syn1neg[t]>; |f| >= 6 skips; g = (label - sigmoid_table[f]) * alpha;
//   work += g * syn1neg[t]; syn1neg[t] += g * h; finally syn0[word_j] += work.
//
// Mapping to the machine: one wavefront owns one walk at a time; a row of d = 64*VPL floats
// is VPL consecutive floats per lane, so a row access is one coalesced wave-wide load or
// store (512 B at d = 128).  The centre word's syn1neg row stays in registers for all of
// its context pairs.  The up-to-8 dot products of a pair are reduced together
// (DPP / ds_swizzle butterflies that halve the value count at each of the first three
// steps), so that lane bitrev3(k) ends up with <h, row_k>, evaluates the sigmoid table and
// the gradient for its own target, and the g's return to all lanes by v_readlane.
// Rows are updated with plain loads/stores, racing with other wavefronts exactly as
// gensim's Hogwild worker threads race with each other.  The path is HBM/L2 gather-scatter
// bound; there is no dense contraction worth an MFMA.
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "n2v_common.h"

#pragma clang fp contract(fast)

namespace {

constexpr int kExpTableSize = 1000;  // gensim EXP_TABLE_SIZE
constexpr float kMaxExp = 6.0f;      // gensim MAX_EXP
__constant__ float c_exp_table[kExpTableSize];

constexpr uint64_t kLcgA = 25214903917ULL, kLcgC = 11ULL, kLcgMask = (1ULL << 48) - 1;

struct SgnsArgs {
    const int32_t* walks;
    const int32_t* lens;
    int64_t n_walks;
    int32_t walk_stride;
    float* syn0;
    float* syn1neg;
    int32_t row_stride;
    int32_t window, negative;
    const uint32_t* sample_int;
    const uint32_t* cum_table;
    const uint32_t* lut;
    int32_t lut_shift;  // 31 - lut_bits
    float alpha0, min_alpha;
    int64_t sent_base, sent_step, sent_total, alpha_batch;
    uint64_t seed, walk_id_base;
    unsigned long long* pair_count;
    unsigned long long* work;    // NULL: static grid stride; else the in-order item counter (reset by the launch)
    int32_t lpad;
    int32_t splits;   // wavefronts per walk (>= 1): split s trains the centres [s*n/S, (s+1)*n/S) of the sentence
    int32_t predraw;  // 1: all negatives of a centre are drawn by the lanes in parallel before its pairs (short launches)
    // span mode (n2v_sgns_train_span): which walks this launch trains is read from device memory, so that a captured
    // launch can be replayed for every merge interval of a pass
    const int64_t* dyn;          // NULL, or {base interval index, sentences of earlier epochs}
    int32_t dyn_sub, dyn_subs;   // this launch is sub-interval dyn_sub of dyn_subs per base interval
    int64_t dyn_n_sub_total, dyn_n_local, dyn_shard_offset;
};

// span mode: sub-interval s = interval * subs + sub of the pass covers the local walks [s*n/k, (s+1)*n/k) (the cut of
// n2v_hip/merge.py:chunk_plan); the schedule arguments follow as in the eager driver (n2v_hip/sgns.py:_train_tsum)
__device__ __forceinline__ void resolve_span(SgnsArgs& a) {
    if (!a.dyn) return;
    const int64_t epoch_base = a.dyn[1];
    const int64_t s = a.dyn[0] * a.dyn_subs + a.dyn_sub;
    const int64_t b = s * a.dyn_n_local / a.dyn_n_sub_total, e = (s + 1) * a.dyn_n_local / a.dyn_n_sub_total;
    a.n_walks = e - b;
    a.walks += b * a.walk_stride;
    if (a.lens) a.lens += b;
    a.sent_base = epoch_base + b * a.sent_step;
    a.walk_id_base = (uint64_t)(epoch_base + a.dyn_shard_offset + b);
}

// x -> x advanced by k steps of the sentence's 48-bit LCG (composition of the affine map by squaring)
__device__ __forceinline__ uint64_t lcg_skip(uint64_t x, uint64_t k) {
    uint64_t cur_m = kLcgA, cur_c = kLcgC, acc_m = 1, acc_c = 0;
    while (k) {
        if (k & 1) { acc_m = (acc_m * cur_m) & kLcgMask; acc_c = (acc_c * cur_m + cur_c) & kLcgMask; }
        cur_c = ((cur_m + 1) * cur_c) & kLcgMask;
        cur_m = (cur_m * cur_m) & kLcgMask;
        k >>= 1;
    }
    return (acc_m * x + acc_c) & kLcgMask;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
__device__ __forceinline__ uint32_t hash32(uint64_t seed, uint64_t walk, uint32_t pos, uint32_t salt) {
    return (uint32_t)(mix64(seed ^ mix64(walk * 0x9E3779B97F4A7C15ULL + (((uint64_t)salt << 32) | pos))) >> 32);
}

__device__ __forceinline__ float xor_dpp1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float xor_dpp2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
}
template <int M>
__device__ __forceinline__ float xor_swz(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (M << 10) | 0x1f));
}

// Reduce 8 per-lane partial sums over the wave at once.  On return lane l holds the total
// of value index 4*(l&1) + 2*((l>>1)&1) + ((l>>2)&1), i.e. value k sits in lane bitrev3(k)
// (and in every lane congruent to it mod 8).
__device__ __forceinline__ float reduce8(const float (&p)[8], int lane) {
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    float q[4], r[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float send = b0 ? p[k] : p[k + 4];
        const float keep = b0 ? p[k + 4] : p[k];
        q[k] = keep + xor_dpp1(send);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float send = b1 ? q[k] : q[k + 2];
        const float keep = b1 ? q[k + 2] : q[k];
        r[k] = keep + xor_dpp2(send);
    }
    float s = (b2 ? r[1] : r[0]) + xor_swz<4>(b2 ? r[0] : r[1]);
    s += xor_swz<8>(s);
    s += xor_swz<16>(s);
    s += __shfl_xor(s, 32);
    return s;
}

__device__ __forceinline__ constexpr int bitrev3(int k) { return ((k & 1) << 2) | (k & 2) | ((k >> 2) & 1); }

// bisect_left(cum_table, r) narrowed by a bucket table: lut[b] = bisect_left(cum_table, b << shift)
__device__ __forceinline__ int32_t draw_target(const uint32_t* __restrict__ cum, const uint32_t* __restrict__ lut,
                                               int shift, uint32_t r) {
    const uint32_t b = r >> shift;
    uint32_t lo = lut[b], hi = lut[b + 1];
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (cum[mid] < r) lo = mid + 1;
        else hi = mid;
    }
    return (int32_t)lo;
}

template <int VPL>
struct Row {
    float v[VPL];
};

// How rows are shared between wavefronts (all of them race, as gensim's Hogwild threads do):
//  kPlain  : plain loads and stores.  Lines live in the issuing XCD's write-back L2, which is
//            not coherent with the other seven: an XCD keeps training on its own copy of a hot
//            row and whole-row write-backs overwrite each other.  Fastest; loses updates.
//  kAgent  : agent-scope (sc1) loads and stores — every access goes to the memory side
//            (Infinity Cache / HBM), so all wavefronts see one copy; a read-modify-write can
//            still lose a concurrent update.
//  kAtomic : agent-scope loads, and every update applied as a float atomic add at the memory
//            side (global_atomic_add_f32, 256 contiguous bytes per wave-instruction): no
//            update is ever lost.  Default.
enum : int { kPlain = 0, kAgent = 1, kAtomic = 2 };

template <int MODE>
__device__ __forceinline__ float2 ld2(const float* p) {
    if constexpr (MODE == kPlain) {
        return *reinterpret_cast<const float2*>(p);
    } else {
        const uint64_t b = __hip_atomic_load(reinterpret_cast<const uint64_t*>(p), __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        return make_float2(__builtin_bit_cast(float, (uint32_t)b), __builtin_bit_cast(float, (uint32_t)(b >> 32)));
    }
}
template <int MODE>
__device__ __forceinline__ void st2(float* p, float x, float y) {
    if constexpr (MODE == kPlain) {
        *reinterpret_cast<float2*>(p) = make_float2(x, y);
    } else {
        const uint64_t b = (uint64_t)__builtin_bit_cast(uint32_t, x) | ((uint64_t)__builtin_bit_cast(uint32_t, y) << 32);
        __hip_atomic_store(reinterpret_cast<uint64_t*>(p), b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Lane layout of a row: VPL <= 2: lane owns VPL consecutive floats; VPL >= 4: 1-KiB chunks of
// the row, 4 floats per lane in each (every wave-instruction touches contiguous bytes).
template <int VPL>
__device__ __forceinline__ int lane_off(int lane) { return VPL <= 2 ? lane * VPL : lane * 4; }

template <int VPL, int MODE>
__device__ __forceinline__ Row<VPL> load_row(const float* base, int64_t row, int stride, int lane) {
    Row<VPL> r;
    if constexpr (MODE == kAtomic) {
        // element i of the lane = float i*64 + lane: each wave-instruction covers 256 contiguous
        // bytes, the shape the memory-side float atomics (add_row) run at full rate for
        const float* q = base + row * stride + lane;
#pragma unroll
        for (int i = 0; i < VPL; ++i) r.v[i] = __hip_atomic_load(q + i * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return r;
    }
    const float* p = base + row * stride + lane_off<VPL>(lane);
    if constexpr (VPL == 1) {
        if constexpr (MODE == kPlain) r.v[0] = *p;
        else r.v[0] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (VPL == 2) {
        const float2 t = ld2<MODE>(p);
        r.v[0] = t.x; r.v[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < VPL; i += 4) {
            if constexpr (MODE == kPlain) {
                const float4 t = *reinterpret_cast<const float4*>(p + i * 64);
                r.v[i] = t.x; r.v[i + 1] = t.y; r.v[i + 2] = t.z; r.v[i + 3] = t.w;
            } else {
                const float2 t0 = ld2<MODE>(p + i * 64), t1 = ld2<MODE>(p + i * 64 + 2);
                r.v[i] = t0.x; r.v[i + 1] = t0.y; r.v[i + 2] = t1.x; r.v[i + 3] = t1.y;
            }
        }
    }
    return r;
}

template <int VPL, int MODE>
__device__ __forceinline__ void store_row(float* base, int64_t row, int stride, int lane, const Row<VPL>& r) {
    float* p = base + row * stride + lane_off<VPL>(lane);
    if constexpr (VPL == 1) {
        if constexpr (MODE == kPlain) *p = r.v[0];
        else __hip_atomic_store(p, r.v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if constexpr (VPL == 2) {
        st2<MODE>(p, r.v[0], r.v[1]);
    } else {
#pragma unroll
        for (int i = 0; i < VPL; i += 4) {
            if constexpr (MODE == kPlain) {
                *reinterpret_cast<float4*>(p + i * 64) = make_float4(r.v[i], r.v[i + 1], r.v[i + 2], r.v[i + 3]);
            } else {
                st2<MODE>(p + i * 64, r.v[i], r.v[i + 1]);
                st2<MODE>(p + i * 64 + 2, r.v[i + 2], r.v[i + 3]);
            }
        }
    }
}

// row += delta, one float atomic per element, at the memory side
template <int VPL>
__device__ __forceinline__ void add_row(float* base, int64_t row, int stride, int lane, const Row<VPL>& d) {
    float* q = base + row * stride + lane;  // same lane layout as load_row<VPL, kAtomic>
#pragma unroll
    for (int i = 0; i < VPL; ++i)
        __hip_atomic_fetch_add(q + i * 64, d.v[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// row += delta in load_row<VPL, kAgent>'s lane layout (the lane's own floats): the centre row of kAgent, see sgns_kernel
template <int VPL>
__device__ __forceinline__ void add_row_packed(float* base, int64_t row, int stride, int lane, const Row<VPL>& d) {
    float* p = base + row * stride + lane_off<VPL>(lane);
    if constexpr (VPL <= 2) {
#pragma unroll
        for (int v = 0; v < VPL; ++v) __hip_atomic_fetch_add(p + v, d.v[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
#pragma unroll
        for (int i = 0; i < VPL; i += 4)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                __hip_atomic_fetch_add(p + i * 64 + v, d.v[i + v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// G = target slots in use per group of 8 (6 when negative == 5: the centre + 5 draws)
// Sentences (items) are handed to the wavefronts IN ORDER by a device counter: every wave then works inside one narrow,
// moving window of the corpus, like the threads of the sequential algorithm's job queue.  With the static grid stride
// (item = wave, wave + n_waves, ...) the waves drift apart — a wave on a fuller CU falls behind by whole percents of the
// corpus — and the link-prediction AUC moved away from the sequential comparator with the grid (399 846
// rows, lossless rows: -0.0023 at 3072 workgroups, -0.0035 at 1561; in order: -0.00003 at every grid).
__device__ __forceinline__ int64_t next_item(unsigned long long* counter, int lane) {
    unsigned long long v = 0;
    if (lane == 0) v = atomicAdd(counter, 1ull);
    const int lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
template <int VPL, int G, int MODE>
// 8 waves per SIMD: the default allocation (106 SGPRs) stops at 7; capped, the kernel fits 78 SGPRs / 61 VGPRs without
// spilling and the agent-row pass gains 1.4 % (10^6-row probe: 9.61 -> 9.74e8 pairs/s)
__attribute__((amdgpu_waves_per_eu(8, 8)))
__global__ void __launch_bounds__(256) sgns_kernel(SgnsArgs a_in) {
    SgnsArgs a = a_in;
    resolve_span(a);
    extern __shared__ int32_t smem[];
    const int lane = threadIdx.x & 63;
    // the wave index is the same in all 64 lanes: tell the compiler, so that everything derived from
    // it (walk id, loop bounds, table sizes) is scalar and loops branch on SCC instead of EXEC
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int32_t* sent = smem + wv * a.lpad;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int my_k = bitrev3(lane & 7);  // which of the 8 reduced values this lane ends up holding
    unsigned long long pairs_done = 0;

    // Work items are (walk, split): with splits == 1 one wavefront owns a walk (gensim's worker owns a sentence); with
    // S > 1 the centres of a sentence are dealt to S wavefronts — the same pairs, the same draws (the sentence's LCG is
    // advanced to each split's first centre in closed form), only the order inside the sentence becomes a race like the
    // one between sentences.  That is what lets a launch of a few hundred walks fill the chip (tiered merges).
    const int S = a.splits;
    const int64_t n_items = a.n_walks * S;
    // (every item comes off the counter, the first one too: a workgroup that only becomes resident late in the pass must not
    // start with the early sentence its index names — measured: -0.0037 at 3072 workgroups on the 400k fixture)
    for (int64_t item = a.work ? next_item(a.work, lane) : (int64_t)blockIdx.x * 4 + wv; item < n_items;
         item = a.work ? next_item(a.work, lane) : item + n_waves) {
        const int64_t wi = S == 1 ? item : item / S;
        const int sp = S == 1 ? 0 : (int)(item - wi * S);
        const int len = a.lens ? a.lens[wi] : a.walk_stride;
        const uint64_t wid = a.walk_id_base + (uint64_t)wi;
        // ---- effective sentence: drop padding and sub-sampled words, keep order
        int n_eff = 0;
        for (int base = 0; base < len; base += 64) {
            const int pos = base + lane;
            bool keep = false;
            int32_t tok = -1;
            if (pos < len) {
                tok = a.walks[wi * a.walk_stride + pos];
                keep = tok >= 0;
                if (keep && a.sample_int) keep = !(a.sample_int[tok] < hash32(a.seed, wid, (uint32_t)pos, 0x5AB));
            }
            const unsigned long long m = __ballot(keep);
            if (keep) sent[n_eff + __popcll(m & ((1ULL << lane) - 1ULL))] = tok;
            n_eff += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- learning rate of this walk (gensim: linear decay, stepped per job)
        const int64_t pushed = a.sent_base + (wi / a.alpha_batch) * a.alpha_batch * a.sent_step;
        float alpha = a.alpha0 - (a.alpha0 - a.min_alpha) * (float)((double)pushed / (double)a.sent_total);
        alpha = fmaxf(alpha, a.min_alpha);

        uint64_t lcg = mix64(a.seed ^ mix64(wid + 0x632BE59BD9B4E019ULL)) & kLcgMask;
        int i_begin = 0, i_end = n_eff;
        if (S > 1) {
            i_begin = (int)((int64_t)sp * n_eff / S);
            i_end = (int)((int64_t)(sp + 1) * n_eff / S);
            // draws of the centres before i_begin: `negative` per (centre, context) pair
            int pairs_before = 0;
            for (int base = 0; base < i_begin; base += 64) {
                const int i = base + lane;
                int np = 0;
                if (i < i_begin) {
                    const int rb = (int)(hash32(a.seed, wid, (uint32_t)i, 0xB17) % (uint32_t)a.window);
                    const int lo = max(0, i - a.window + rb), hi = min(n_eff, i + a.window + 1 - rb);
                    np = hi - lo > 1 ? hi - lo - 1 : 0;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) np += __shfl_xor(np, o, 64);
                pairs_before += np;
            }
            lcg = lcg_skip(lcg, (uint64_t)pairs_before * (uint64_t)a.negative);
        }

        for (int i = i_begin; i < i_end; ++i) {
            const int32_t ci = __builtin_amdgcn_readfirstlane(sent[i]);
            const int rb = (int)(hash32(a.seed, wid, (uint32_t)i, 0xB17) % (uint32_t)a.window);
            const int lo = max(0, i - a.window + rb), hi = min(n_eff, i + a.window + 1 - rb);
            if (hi - lo <= 1) continue;
            Row<VPL> c = load_row<VPL, MODE>(a.syn1neg, ci, a.row_stride, lane);
            Row<VPL> cd;  // kAtomic: this wave's accumulated change of the centre row
#pragma unroll
            for (int v = 0; v < VPL; ++v) cd.v[v] = 0.f;
            // A wave's pairs are a chain of dependent loads, and the look-up of the negatives (bucket index, then a
            // bisect of the cumulative table) is two to three links of it per pair — visible even in full-size launches
            // at 7 waves per SIMD.  With predraw the lanes make ALL draws of the centre at once — draw number d
            // of the centre uses the walk's LCG advanced d times, exactly the state the pair-by-pair path reaches —
            // and a pair fetches its targets from the lanes that hold them.
            const int nd = (hi - lo - 1) * a.negative;
            const bool pre = a.predraw && nd <= 128;
            int32_t drawn0 = -1, drawn1 = -1;   // draws 0..63 and 64..127 of this centre, one per lane
            if (pre) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int d = half * 64 + lane;
                    int32_t t = -1;
                    if (d < nd) {
                        const uint64_t s = lcg_skip(lcg, (uint64_t)d);
                        t = draw_target(a.cum_table, a.lut, a.lut_shift, (uint32_t)((s >> 16) % 2147483647ULL));
                        if (t == ci) t = -1;  // `if target_index == word_index: continue`
                    }
                    if (half == 0) drawn0 = t;
                    else drawn1 = t;
                }
            }
            int pidx = 0;  // number of this pair among the centre's pairs
            for (int j = lo; j < hi; ++j) {
                if (j == i) continue;
                const int32_t xj = __builtin_amdgcn_readfirstlane(sent[j]);
                Row<VPL> h = load_row<VPL, MODE>(a.syn0, xj, a.row_stride, lane);
                Row<VPL> work;
#pragma unroll
                for (int v = 0; v < VPL; ++v) work.v[v] = 0.f;
                // targets are processed 8 at a time: slot 0 of the first group is the centre word
                for (int t0 = 0; t0 < a.negative + 1; t0 += 8) {
                    // lane k (k < 8) draws the target of slot k of this group
                    int32_t my_t = -1;
                    if (pre) {   // negative <= 7: one group, lane k in [1, negative] holds target k
                        const int d = min(max(pidx * a.negative + lane - 1, 0), 127);
                        const int v0 = __builtin_amdgcn_ds_bpermute((d & 63) << 2, drawn0);
                        const int v1 = __builtin_amdgcn_ds_bpermute((d & 63) << 2, drawn1);
                        if (lane >= 1 && lane <= a.negative) my_t = d < 64 ? v0 : v1;
                    } else {
                        const int tk = t0 + lane;  // target number: 0 = positive, d >= 1 = d-th negative
                        if (lane < 8 && tk >= 1 && tk <= a.negative) {
                            uint64_t s = lcg;  // state of the first draw of this group
                            for (int d = max(t0, 1); d < tk; ++d) s = (s * kLcgA + kLcgC) & kLcgMask;
                            const uint32_t r = (uint32_t)((s >> 16) % 2147483647ULL);
                            my_t = draw_target(a.cum_table, a.lut, a.lut_shift, r);
                            if (my_t == ci) my_t = -1;  // `if target_index == word_index: continue`
                        }
                    }
                    int32_t tgt[G];
                    Row<VPL> n[G];
                    float p[8];
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        tgt[k] = __builtin_amdgcn_readlane(my_t, k);
                        if (k == 0 && t0 == 0) tgt[k] = ci;
                    }
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        if (k == 0 && t0 == 0) {
                            n[k] = c;
                        } else if (tgt[k] >= 0) {
                            n[k] = load_row<VPL, MODE>(a.syn1neg, tgt[k], a.row_stride, lane);
                        } else {
#pragma unroll
                            for (int v = 0; v < VPL; ++v) n[k].v[v] = 0.f;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float acc = 0.f;
                        if (k < G) {
#pragma unroll
                            for (int v = 0; v < VPL; ++v) acc = fmaf(h.v[v], n[k].v[v], acc);
                        }
                        p[k] = acc;
                    }
                    const float f = reduce8(p, lane);
                    // this lane's own target: sigmoid table, gradient
                    float g = 0.f;
                    if (f > -kMaxExp && f < kMaxExp) {
                        const float sig = c_exp_table[(int)((f + kMaxExp) * (float)(kExpTableSize / (int)kMaxExp / 2))];
                        const float label = (my_k == 0 && t0 == 0) ? 1.f : 0.f;
                        g = (label - sig) * alpha;
                    }
#pragma unroll
                    for (int k = 0; k < G; ++k) {
                        if (tgt[k] < 0) continue;
                        const float gk = __builtin_bit_cast(
                            float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g), bitrev3(k)));
                        if (gk == 0.f) continue;  // |f| >= MAX_EXP: no update at all
                        Row<VPL> dn;
#pragma unroll
                        for (int v = 0; v < VPL; ++v) {
                            work.v[v] = fmaf(gk, n[k].v[v], work.v[v]);
                            dn.v[v] = gk * h.v[v];
                            n[k].v[v] += dn.v[v];
                        }
                        if (k == 0 && t0 == 0) {
                            c = n[k];
#pragma unroll
                            for (int v = 0; v < VPL; ++v) cd.v[v] += dn.v[v];
                        } else if constexpr (MODE == kAtomic) {
                            add_row<VPL>(a.syn1neg, tgt[k], a.row_stride, lane, dn);
                        } else {
                            store_row<VPL, MODE>(a.syn1neg, tgt[k], a.row_stride, lane, n[k]);
                        }
                    }
                    // advance the walk's LCG past this group's negatives
                    const int used = min(a.negative, t0 + 7) - max(t0, 1) + 1;
                    for (int d = 0; d < used; ++d) lcg = (lcg * kLcgA + kLcgC) & kLcgMask;
                }
                if constexpr (MODE == kAtomic) {
                    add_row<VPL>(a.syn0, xj, a.row_stride, lane, work);
#ifdef N2V_SGNS_LAB_CONTEXT_ADD
                } else if constexpr (MODE == kAgent) {
                    add_row_packed<VPL>(a.syn0, xj, a.row_stride, lane, work);
#endif
                } else {
#pragma unroll
                    for (int v = 0; v < VPL; ++v) h.v[v] += work.v[v];
                    store_row<VPL, MODE>(a.syn0, xj, a.row_stride, lane, h);
                }
                ++pidx;
                ++pairs_done;
            }
            // The centre row sits in registers for the whole window (~20 pairs, tens of microseconds): written back whole it
            // would erase every update other waves made to it meanwhile — by far the longest exposure of any row.  kAgent
            // therefore ADDS this wave's accumulated change, like kAtomic (one atomic row per centre: < 1 % of the row
            // updates); the context and negative rows, held for one pair, keep their whole-row stores.
            if constexpr (MODE == kAtomic) add_row<VPL>(a.syn1neg, ci, a.row_stride, lane, cd);
            else if constexpr (MODE == kAgent) add_row_packed<VPL>(a.syn1neg, ci, a.row_stride, lane, cd);
            else store_row<VPL, MODE>(a.syn1neg, ci, a.row_stride, lane, c);
        }
        __builtin_amdgcn_wave_barrier();  // LDS sentence is reused by the next walk
    }
    if (a.pair_count && lane == 0 && pairs_done) atomicAdd(a.pair_count, pairs_done);
}

// Opt-in variant (N2V_SGNS_SHARE_NEGATIVES): the `negative` draws are made once per CENTRE word
// and shared by all of its context pairs (the scheme of Ji et al., "Parallelizing Word2Vec in
// Shared and Distributed Memory", 2016) instead of once per pair as gensim does.  The target
// rows then stay in registers for the whole window and are written back once per centre, so a
// pair touches memory only for its context row: ~0.8 KB instead of 3.1 KB of atomic traffic.
// Same update rule per (pair, target); different (correlated) negative samples.  negative <= 7.
template <int VPL, int MODE>
__global__ void __launch_bounds__(256) sgns_shared_kernel(SgnsArgs a_in) {
    SgnsArgs a = a_in;
    resolve_span(a);
    extern __shared__ int32_t smem[];
    const int lane = threadIdx.x & 63;
    // the wave index is the same in all 64 lanes: tell the compiler, so that everything derived from
    // it (walk id, loop bounds, table sizes) is scalar and loops branch on SCC instead of EXEC
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int32_t* sent = smem + wv * a.lpad;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const int my_k = bitrev3(lane & 7);
    unsigned long long pairs_done = 0;

    for (int64_t wi = (int64_t)blockIdx.x * 4 + wv; wi < a.n_walks; wi += n_waves) {
        const int len = a.lens ? a.lens[wi] : a.walk_stride;
        const uint64_t wid = a.walk_id_base + (uint64_t)wi;
        int n_eff = 0;
        for (int base = 0; base < len; base += 64) {
            const int pos = base + lane;
            bool keep = false;
            int32_t tok = -1;
            if (pos < len) {
                tok = a.walks[wi * a.walk_stride + pos];
                keep = tok >= 0;
                if (keep && a.sample_int) keep = !(a.sample_int[tok] < hash32(a.seed, wid, (uint32_t)pos, 0x5AB));
            }
            const unsigned long long m = __ballot(keep);
            if (keep) sent[n_eff + __popcll(m & ((1ULL << lane) - 1ULL))] = tok;
            n_eff += __popcll(m);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t pushed = a.sent_base + (wi / a.alpha_batch) * a.alpha_batch * a.sent_step;
        float alpha = a.alpha0 - (a.alpha0 - a.min_alpha) * (float)((double)pushed / (double)a.sent_total);
        alpha = fmaxf(alpha, a.min_alpha);
        uint64_t lcg = mix64(a.seed ^ mix64(wid + 0x632BE59BD9B4E019ULL)) & kLcgMask;

        for (int i = 0; i < n_eff; ++i) {
            const int32_t ci = __builtin_amdgcn_readfirstlane(sent[i]);
            const int rb = (int)(hash32(a.seed, wid, (uint32_t)i, 0xB17) % (uint32_t)a.window);
            const int lo = max(0, i - a.window + rb), hi = min(n_eff, i + a.window + 1 - rb);
            if (hi - lo <= 1) continue;
            // targets of this centre: slot 0 = the centre itself, slots 1..negative = one draw each
            int32_t my_t = -1;
            if (lane >= 1 && lane <= a.negative) {
                uint64_t s = lcg;
                for (int d = 1; d < lane; ++d) s = (s * kLcgA + kLcgC) & kLcgMask;
                my_t = draw_target(a.cum_table, a.lut, a.lut_shift, (uint32_t)((s >> 16) % 2147483647ULL));
                if (my_t == ci) my_t = -1;
            }
            for (int d = 0; d < a.negative; ++d) lcg = (lcg * kLcgA + kLcgC) & kLcgMask;
            int32_t tgt[8];
            Row<VPL> n[8], dn[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                tgt[k] = (k == 0) ? ci : __builtin_amdgcn_readlane(my_t, k);
                // a target drawn twice is trained once (its second copy would race with the first)
#pragma unroll
                for (int k2 = 1; k2 < k; ++k2)
                    if (tgt[k] >= 0 && tgt[k] == tgt[k2]) tgt[k] = -1;
                if (tgt[k] >= 0) n[k] = load_row<VPL, MODE>(a.syn1neg, tgt[k], a.row_stride, lane);
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    if (tgt[k] < 0) n[k].v[v] = 0.f;
                    dn[k].v[v] = 0.f;
                }
            }
            for (int j = lo; j < hi; ++j) {
                if (j == i) continue;
                const int32_t xj = __builtin_amdgcn_readfirstlane(sent[j]);
                Row<VPL> h = load_row<VPL, MODE>(a.syn0, xj, a.row_stride, lane);
                float p[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float acc = 0.f;
#pragma unroll
                    for (int v = 0; v < VPL; ++v) acc = fmaf(h.v[v], n[k].v[v], acc);
                    p[k] = acc;
                }
                const float f = reduce8(p, lane);
                float g = 0.f;
                if (f > -kMaxExp && f < kMaxExp) {
                    const float sig = c_exp_table[(int)((f + kMaxExp) * (float)(kExpTableSize / (int)kMaxExp / 2))];
                    g = ((my_k == 0 ? 1.f : 0.f) - sig) * alpha;
                }
                Row<VPL> work;
#pragma unroll
                for (int v = 0; v < VPL; ++v) work.v[v] = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (tgt[k] < 0) continue;
                    const float gk = __builtin_bit_cast(
                        float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g), bitrev3(k)));
                    if (gk == 0.f) continue;
#pragma unroll
                    for (int v = 0; v < VPL; ++v) {
                        work.v[v] = fmaf(gk, n[k].v[v], work.v[v]);
                        const float d = gk * h.v[v];
                        n[k].v[v] += d;
                        dn[k].v[v] += d;
                    }
                }
                if constexpr (MODE == kAtomic) {
                    add_row<VPL>(a.syn0, xj, a.row_stride, lane, work);
#ifdef N2V_SGNS_LAB_CONTEXT_ADD
                } else if constexpr (MODE == kAgent) {
                    add_row_packed<VPL>(a.syn0, xj, a.row_stride, lane, work);
#endif
                } else {
#pragma unroll
                    for (int v = 0; v < VPL; ++v) h.v[v] += work.v[v];
                    store_row<VPL, MODE>(a.syn0, xj, a.row_stride, lane, h);
                }
                ++pairs_done;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (tgt[k] < 0) continue;
                if constexpr (MODE == kAtomic) add_row<VPL>(a.syn1neg, tgt[k], a.row_stride, lane, dn[k]);
                else store_row<VPL, MODE>(a.syn1neg, tgt[k], a.row_stride, lane, n[k]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (a.pair_count && lane == 0 && pairs_done) atomicAdd(a.pair_count, pairs_done);
}

// syn0 ~ U(-0.5/d, 0.5/d), syn1neg = 0 (gensim reset_weights); one Philox call per 4 floats,
// keyed by the seed and counted by (row, column block) so a row does not depend on n_words.
__global__ void __launch_bounds__(256)
sgns_init_kernel(float* syn0, float* syn1neg, int64_t n_words, int32_t dim, int32_t stride, uint64_t seed) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;  // one thread per 4 columns
    const int blocks_per_row = stride / 4;
    const int64_t row = idx / blocks_per_row;
    const int cb = (int)(idx - row * blocks_per_row);
    if (row >= n_words) return;
    uint32_t c0 = (uint32_t)row, c1 = (uint32_t)(row >> 32), c2 = (uint32_t)cb, c3 = 0x5EEDu;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint32_t rr[4] = {c0, c1, c2, c3};
    float4 o, z = make_float4(0.f, 0.f, 0.f, 0.f);
    float* po = &o.x;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int colx = cb * 4 + t;
        const float u = (float)(rr[t] >> 8) * (1.0f / 16777216.0f);  // [0,1), 24 bits
        po[t] = colx < dim ? (u - 0.5f) / (float)dim : 0.f;
    }
    *reinterpret_cast<float4*>(syn0 + row * stride + cb * 4) = o;
    *reinterpret_cast<float4*>(syn1neg + row * stride + cb * 4) = z;
}

__global__ void __launch_bounds__(256)
neg_lut_kernel(const uint32_t* __restrict__ cum, int64_t n_words, int shift, int64_t n_buckets, uint32_t* __restrict__ lut) {
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b > n_buckets) return;
    const uint64_t key = (uint64_t)b << shift;
    int64_t lo = 0, hi = n_words;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((uint64_t)cum[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    lut[b] = (uint32_t)lo;
}

float host_exp_table[kExpTableSize];
bool host_exp_ready = false;

void fill_exp_table() {
    if (host_exp_ready) return;
    for (int i = 0; i < kExpTableSize; ++i) {
        // gensim word2vec_inner.pyx init(): EXP_TABLE[i] = exp((i / 1000 * 2 - 1) * 6); e / (e + 1), float32
        const float x = ((float)i / (float)kExpTableSize * 2.0f - 1.0f) * kMaxExp;
        const float e = (float)std::exp((double)x);
        host_exp_table[i] = (float)(e / (e + 1.0f));
    }
    host_exp_ready = true;
}

// The sigmoid table reaches each device's constant memory once per process (a blocking copy the first time a device
// trains): re-uploading it with every launch cost a 4-KB copy kernel and ~20 us of host time per launch, which the
// thousands of short launches of the tiered merges pay in full.
std::mutex exp_upload_mutex;
bool exp_uploaded[64] = {};

int upload_exp_table() {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) return n2v::fail(N2V_ERR_HIP, "n2v_sgns_train: hipGetDevice: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(exp_upload_mutex);
    if (exp_uploaded[dev]) return N2V_OK;
    fill_exp_table();
    e = hipMemcpyToSymbol(HIP_SYMBOL(c_exp_table), host_exp_table, sizeof(host_exp_table), 0, hipMemcpyHostToDevice);
    if (e != hipSuccess) return n2v::fail(N2V_ERR_HIP, "n2v_sgns_train: exp table upload: %s", hipGetErrorString(e));
    exp_uploaded[dev] = true;
    return N2V_OK;
}

}  // namespace

extern "C" int n2v_build_neg_lut(const uint32_t* cum_table, int64_t n_words, int32_t lut_bits, uint32_t* lut,
                                 void* stream) {
    if (!cum_table || !lut || n_words <= 0 || n_words >= ((int64_t)1 << 32) || lut_bits < 1 || lut_bits > 24)
        return n2v::fail(N2V_ERR_INVALID, "n2v_build_neg_lut: bad argument (n_words %lld, lut_bits %d)",
                         (long long)n_words, (int)lut_bits);
    const int64_t nb = (int64_t)1 << lut_bits;
    hipLaunchKernelGGL(neg_lut_kernel, dim3(n2v::grid_for(nb + 1, 256)), dim3(256), 0, (hipStream_t)stream, cum_table,
                       n_words, 31 - lut_bits, nb, lut);
    return n2v::check_launch("n2v_build_neg_lut");
}

extern "C" int n2v_sgns_init(float* syn0, float* syn1neg, int64_t n_words, int32_t dim, int32_t row_stride,
                             uint64_t seed, void* stream) {
    if (!syn0 || !syn1neg || n_words < 0 || dim < 1 || row_stride < dim || (row_stride % 4) != 0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_init: bad argument");
    if (n_words == 0) return N2V_OK;
    const int64_t threads = n_words * (row_stride / 4);
    hipLaunchKernelGGL(sgns_init_kernel, dim3(n2v::grid_for(threads, 256)), dim3(256), 0, (hipStream_t)stream, syn0,
                       syn1neg, n_words, dim, row_stride, seed);
    return n2v::check_launch("n2v_sgns_init");
}

namespace {
// predraw (parallel draws of a centre's negatives): on whenever negative <= 7 (one target group).  Measured on C3's
// walks: full-size launches 7.98e8 -> 8.85e8 pairs/s, one wavefront per walk (latency-bound) 3.49 -> 2.67 us per pair,
// the 83-walk launches of the tiered merges 145 -> 128 us.  N2V_SGNS_PREDRAW=0 switches it off (A/B timing, and the
// test that both paths train the same bits).
int predraw_mode(int walk_splits, int negative) {
    (void)walk_splits;
    if (negative < 1 || negative > 7) return 0;
    const char* e = getenv("N2V_SGNS_PREDRAW");
    return (e && e[0] == '0') ? 0 : 1;
}
}  // namespace

namespace {
struct SpanSpec {                // n2v_sgns_train_span; dyn == NULL: an ordinary launch
    const int64_t* dyn;
    int32_t sub, subs;
    int64_t n_sub_total, n_local, shard_offset;
};

// CUs of the current device (MI355X: 256), asked once
static int64_t n2v_cu_count() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
        else cus = 256;
    }
    return cus;
}

// the default grid's workgroup count (rules (1) and (2) in sgns_launch)
static int64_t default_grid(int64_t n_words, int32_t update_mode) {
    const int64_t cus = n2v_cu_count();
    int64_t cap = 3072;
    if (cap > n_words / 256) cap = n_words / 256 > 16 ? n_words / 256 : 16;
    // store-based rows (agent, plain) need ~4 workgroups per CU for their pair rate (131 019 rows: 256 workgroups 5.0e8
    // pairs/s, 1024 1.09e9) and, with the in-order hand-out, hold the band there (+0.0002 at 1024 ... 2048 workgroups)
    if (update_mode != kAtomic && cap < 4 * cus) cap = 4 * cus;
    if (cap > cus) cap -= cap % cus;
    return cap;
}

int sgns_launch(const char* who, const int32_t* walks, const int32_t* lens, int64_t n_walks, int32_t walk_stride,
                float* syn0, float* syn1neg, int64_t n_words, int32_t dim, int32_t row_stride,
                int32_t window, int32_t negative, const uint32_t* sample_int,
                const uint32_t* cum_table, const uint32_t* lut, int32_t lut_bits, float alpha,
                float min_alpha, int64_t sentences_base, int64_t sentences_step,
                int64_t sentences_total, int64_t alpha_batch,
                uint64_t seed, uint64_t walk_id_base, unsigned long long* pair_count,
                int32_t update_mode, int32_t max_blocks, int32_t walk_splits, const SpanSpec& span,
                unsigned long long* work_counter, void* stream) {
    if (walk_splits < 1 || walk_splits > walk_stride)
        return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: walk_splits %d outside [1, %d]", (int)walk_splits, (int)walk_stride);
    if (n_walks < 0 || walk_stride < 1 || n_words < 1 || dim < 1 || window < 1 || negative < 0 || negative > 64)
        return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: bad size (walks %lld x %d, words %lld, dim %d, window %d, negative %d)",
                         (long long)n_walks, (int)walk_stride, (long long)n_words, (int)dim, (int)window, (int)negative);
    if (n_walks == 0) return N2V_OK;
    if (!walks || !syn0 || !syn1neg || (negative > 0 && (!cum_table || !lut)))
        return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: null pointer");
    if (row_stride < dim || (row_stride % 64) != 0 || row_stride > 512)
        return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: row_stride %d must be a multiple of 64 in [dim, 512]",
                         (int)row_stride);
    if (lut_bits < 1 || lut_bits > 24) return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: lut_bits %d", (int)lut_bits);
    const bool share = (update_mode & N2V_SGNS_SHARE_NEGATIVES) != 0;
    const bool unchecked = (update_mode & N2V_SGNS_UNCHECKED) != 0;
    update_mode &= ~(N2V_SGNS_SHARE_NEGATIVES | N2V_SGNS_UNCHECKED);
    if (update_mode < kPlain || update_mode > kAtomic)
        return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: update_mode %d", (int)update_mode);
    if (share && negative > 7) return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: shared negatives need negative <= 7");
    if (share && walk_splits != 1) return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: shared negatives need walk_splits == 1");
    // the centres of ONE sentence dealt to several wavefronts update the same context rows at the same instant: with
    // non-atomic read-modify-write rows that is where updates are lost most — never scored against the comparator
    if (walk_splits > 1 && update_mode != kAtomic && !unchecked)
        return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: walk_splits > 1 needs N2V_SGNS_ATOMIC (or N2V_SGNS_UNCHECKED)");
    if (sentences_total < 1 || alpha_batch < 1 || sentences_step < 1) return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: bad schedule");
    hipStream_t st = (hipStream_t)stream;
    if (int rc = upload_exp_table()) return rc;

    SgnsArgs a;
    a.walks = walks; a.lens = lens; a.n_walks = n_walks; a.walk_stride = walk_stride;
    a.syn0 = syn0; a.syn1neg = syn1neg; a.row_stride = row_stride;
    a.window = window; a.negative = negative; a.sample_int = sample_int;
    a.cum_table = cum_table; a.lut = lut; a.lut_shift = 31 - lut_bits;
    a.alpha0 = alpha; a.min_alpha = min_alpha;
    a.work = share ? nullptr : work_counter;   // (the shared-negatives kernel keeps the static stride)
    a.sent_base = sentences_base; a.sent_step = sentences_step; a.sent_total = sentences_total;
    a.alpha_batch = alpha_batch;
    a.seed = seed; a.walk_id_base = walk_id_base; a.pair_count = pair_count;
    a.lpad = (walk_stride + 63) & ~63;
    a.splits = walk_splits;
    a.predraw = predraw_mode(walk_splits, negative);
    a.dyn = span.dyn; a.dyn_sub = span.sub; a.dyn_subs = span.subs;
    a.dyn_n_sub_total = span.n_sub_total; a.dyn_n_local = span.n_local; a.dyn_shard_offset = span.shard_offset;
    const size_t shmem = (size_t)4 * a.lpad * sizeof(int32_t);
    if (shmem > 64 * 1024) return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: walk_stride %d too long", (int)walk_stride);
    int64_t blocks = (n_walks * walk_splits + 3) / 4;
    // default grid: 256 CUs x 12 workgroups of 4 waves (measured on C3: 2048 blocks 6.9e8 pairs/s, 3072 8.2e8, 4096
    // 8.0e8, 6144 8.3e8; at 8 waves per SIMD 8 of the 12 are resident at a time and the others follow as slots free up —
    // harmless with the in-order hand-out, and the reason the static stride lost the band at large grids: a workgroup
    // that starts late trains its whole strided share of the corpus after everybody else) — but
    //  (1) never more than one wavefront per 64 vocabulary rows: the racing waves read each other's rows stale, and
    //      the link-prediction AUC moves away from the sequential algorithm's in proportion to waves in flight per row.
    //      Against the sequential comparator (tests/probes/grid_band_probe.py, profiles/r03/logs), 131 019-row hub graph:
    //      3072 workgroups -0.0022 (atomic) / -0.0013 (agent), 1024 -0.0005 / -0.0008, 512 -0.0002 / -0.0002, 256 +0.0001 /
    //      -0.0001; 399 846 rows: 3072 -0.0023 / -0.0001, 1536 -0.0016 / -0.0005, 768 -0.0006 / -0.0002;
    //  (2) a whole number of workgroups per CU once there is more than one: with 6 workgroups on most CUs and 7 on a few,
    //      the waves of the fuller CUs fall behind, the pass takes 26 % longer and the AUC drops by 0.004 (399 846 rows:
    //      grids 1560 / 1561 / 1562 / 1600 -0.0041 ... -0.0042 in 4.35 s, 1536 -0.0005 in 3.44 s —
    //      tests/probes/grid_resonance_probe.py).  C3 (10^6 rows) keeps its 3072 = 12 x 256.
    // Both were measured with the static grid stride; with the in-order hand-out (next_item) the AUC no longer depends on
    // the grid (lossless rows: within 3e-5 of the comparator from 768 to 3072 workgroups).  The rules stay: they are what
    // the test suite validated, and they cost no speed.
    const int64_t cap = max_blocks > 0 ? max_blocks : default_grid(n_words, update_mode);
    if (blocks > cap) blocks = cap;
    const dim3 grid((unsigned)blocks), block(256);
    // a launch in which no wave gets a second item needs no hand-out (the replicas' short launches: thousands per pass,
    // and 6 640 waves asking one address for "nothing left" cost 90 us each time)
    if (n_walks * walk_splits <= blocks * 4) a.work = nullptr;
    if (a.work && hipMemsetAsync(a.work, 0, sizeof(unsigned long long), st) != hipSuccess)
        return n2v::fail(N2V_ERR_HIP, "%s: resetting the work counter failed", who);
#define N2V_SGNS_LAUNCH_M(V, M)                                                            \
    if (share) hipLaunchKernelGGL((sgns_shared_kernel<V, M>), grid, block, shmem, st, a);     \
    else if (negative <= 5) hipLaunchKernelGGL((sgns_kernel<V, 6, M>), grid, block, shmem, st, a); \
    else hipLaunchKernelGGL((sgns_kernel<V, 8, M>), grid, block, shmem, st, a)
#define N2V_SGNS_LAUNCH(V)                                   \
    if (update_mode == kPlain) { N2V_SGNS_LAUNCH_M(V, kPlain); }        \
    else if (update_mode == kAgent) { N2V_SGNS_LAUNCH_M(V, kAgent); }   \
    else { N2V_SGNS_LAUNCH_M(V, kAtomic); }
    switch (row_stride / 64) {
        case 1: N2V_SGNS_LAUNCH(1); break;
        case 2: N2V_SGNS_LAUNCH(2); break;
        case 4: N2V_SGNS_LAUNCH(4); break;
        case 8: N2V_SGNS_LAUNCH(8); break;
        default:
            return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train: row_stride %d must be 64, 128, 256 or 512", (int)row_stride);
    }
    return n2v::check_launch(who);
}
}  // namespace

extern "C" int n2v_sgns_train(const int32_t* walks, const int32_t* lens, int64_t n_walks, int32_t walk_stride,
                              float* syn0, float* syn1neg, int64_t n_words, int32_t dim, int32_t row_stride,
                              int32_t window, int32_t negative, const uint32_t* sample_int,
                              const uint32_t* cum_table, const uint32_t* lut, int32_t lut_bits, float alpha,
                              float min_alpha, int64_t sentences_base, int64_t sentences_step,
                              int64_t sentences_total, int64_t alpha_batch,
                              uint64_t seed, uint64_t walk_id_base, unsigned long long* pair_count,
                              int32_t update_mode, int32_t max_blocks, int32_t walk_splits,
                              unsigned long long* work_counter, void* stream) {
    return sgns_launch("n2v_sgns_train", walks, lens, n_walks, walk_stride, syn0, syn1neg, n_words, dim, row_stride, window,
                       negative, sample_int, cum_table, lut, lut_bits, alpha, min_alpha, sentences_base, sentences_step,
                       sentences_total, alpha_batch, seed, walk_id_base, pair_count, update_mode, max_blocks, walk_splits,
                       SpanSpec{nullptr, 0, 1, 1, 0, 0}, work_counter, stream);
}

extern "C" int n2v_sgns_train_span(const int32_t* walks, const int32_t* lens, int64_t n_local, int32_t walk_stride,
                                   float* syn0, float* syn1neg, int64_t n_words, int32_t dim, int32_t row_stride,
                                   int32_t window, int32_t negative, const uint32_t* sample_int,
                                   const uint32_t* cum_table, const uint32_t* lut, int32_t lut_bits, float alpha,
                                   float min_alpha, int64_t sentences_step, int64_t sentences_total, int64_t alpha_batch,
                                   uint64_t seed, unsigned long long* pair_count, int32_t update_mode, int32_t max_blocks,
                                   int32_t walk_splits, const int64_t* interval_state, int32_t sub_index,
                                   int32_t subs_per_interval, int64_t n_sub_total, int64_t shard_offset,
                                   unsigned long long* work_counter, void* stream) {
    if (!interval_state || subs_per_interval < 1 || sub_index < 0 || sub_index >= subs_per_interval || n_sub_total < subs_per_interval ||
        (n_sub_total % subs_per_interval) != 0 || n_local < 0 || shard_offset < 0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_sgns_train_span: bad span (sub %d of %d, %lld sub-intervals, %lld walks)",
                         (int)sub_index, (int)subs_per_interval, (long long)n_sub_total, (long long)n_local);
    // the launch is sized for the longest sub-interval (they differ by at most one walk); the kernel reads its own range
    const int64_t max_walks = (n_local + n_sub_total - 1) / n_sub_total;
    if (max_walks == 0) return N2V_OK;
    return sgns_launch("n2v_sgns_train_span", walks, lens, max_walks, walk_stride, syn0, syn1neg, n_words, dim, row_stride, window,
                       negative, sample_int, cum_table, lut, lut_bits, alpha, min_alpha, 0, sentences_step, sentences_total,
                       alpha_batch, seed, 0, pair_count, update_mode, max_blocks, walk_splits,
                       SpanSpec{interval_state, sub_index, subs_per_interval, n_sub_total, n_local, shard_offset}, work_counter, stream);
}

extern "C" int32_t n2v_sgns_default_blocks(int64_t n_words, int32_t update_mode) {
    return (int32_t)default_grid(n_words < 0 ? 0 : n_words, update_mode & 3);
}
