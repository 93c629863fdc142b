// On-the-fly 2nd-order walk — gfx950 (MI355X).
//
// Replaces Graph.node2vec_walk_on_the_fly / simulate_walks_on_the_fly of the reference
// (src/node2vec.py:13-53,97-111): the alias table of (prev, cur) is rebuilt at every step
// instead of being read from the Σdeg² tables of preprocess_transition_probs — the reference's
// own answer to graphs whose edge tables do not fit in memory.  The result is identical to
// the table-driven walk (same table bits, same two uniforms per step), so walks are
// bit-identical to n2v_walk / the reference under the same uniforms.
//
// One wavefront owns one walk.  Per step the wave
//   1. gathers cur's row (ids, weights) cooperatively and classifies every neighbour
//      (== prev: w/p; has_edge(nbr, prev): w; else w/q — src/node2vec.py:142-148) in parallel,
//   2. runs the two inherently serial pieces in the reference's order — the left-to-right fp64 sum (:149) and
//      Vose's stack pairing (:259-268) — fed from registers (n2v_wave_table.h, shared with the table builder),
//   3. normalises (divide, then multiply by K) in parallel again, and draws.
// The table lives in the wave's slice of LDS while K <= kLdsSlots (256) and in a per-wave global
// scratch row of max_degree slots otherwise.  Serial chains of many waves interleave on a
// SIMD, so throughput comes from occupancy; -ffp-contract=off keeps every rounding separate.
#include "n2v_common.h"
#include "n2v_wave_table.h"

namespace {

#ifndef N2V_OTF_LDS_SLOTS
#define N2V_OTF_LDS_SLOTS 256   /* 6 KiB per wave -> 6 workgroups per CU: C3 2.1 -> 2.6e8 steps/s vs 512 slots / 4 workgroups; 128: 2.5e8 */
#endif
constexpr int kLdsSlots = N2V_OTF_LDS_SLOTS;  // 16 B per slot and wave, plus the feed and row cache of n2v_wave_table.h
constexpr int kOtfRow = 256;   // two row buffers per wave: the row of `prev` (searched) and the row of `cur` (it is the next step's `prev`)
constexpr int kLdsPerWg = 4 * (kLdsSlots * 16 + n2v::kFeed * 8 + 2 * kOtfRow * 4);
constexpr int kWgPerCu = (160 * 1024 / kLdsPerWg) < 8 ? (160 * 1024 / kLdsPerWg) : 8;

using n2v::uni;
using n2v::uni64;

struct OtfArgs {
    n2v::RowCtx g;      // CSR, p, q, symmetric
    const int32_t* starts;
    int64_t n_starts, pos_begin, pos_count, round_begin, n_local;
    int32_t L;
    int32_t rng_mode;
    const double* uniforms;
    const int64_t* walk_uoff;
    uint64_t seed;
    n2v_alias_slot* scratch;  // [n_waves][max_degree]
    int64_t max_degree;
    // hybrid (n2v_walk_hybrid): fat tables of a subset of the CSR entries; a record's table index N2V_NO_TABLE = rebuild
    const n2v_fat_slot* node_fat;
    const n2v_fat_slot* fat;
    const n2v_edge_rec* recs;
    int32_t* walks;
    int32_t* lens;
    int32_t* status;
    // draw before building (n2v_wave_table.h): 1 = after the weights and their sum, 2 = dyadic counting first
    int32_t draw_first;
    int32_t exact_sum;        // draw_first 2: 1/p and 1/q dyadic — the weights' sum is a count
    double wp, wq;            // 1/p, 1/q
};

// The per-wave state of the step that has no stored table: LDS table window, global scratch row, the sum's feed and the
// two row buffers (the row of `prev`, searched by has_edge; the row of `cur`, staged because it is the next step's prev).
struct OtfWave {
    n2v_alias_slot* Tl;
    n2v_alias_slot* Tg;
    n2v::WaveScratch ws;
    int32_t* rows2;
    int pb;                 // which of the two row buffers holds the row of `cached_node`
    int32_t cached_node;
    int cached_n;
    int32_t picked_node;    // WANT_NODE: col[base + pick]

    // alias_draw (:277-281) on the table of the step prev -> cur (row at `base`, K neighbours) for the slot kk = int(u1*K)
    // and the second uniform u2: the picked slot, or -1 when the weights sum to 0 (:150).  All 64 lanes call it with
    // wave-uniform arguments.  keep_row: cur is the next call's prev (one walk per wave) — stage its row for that call.
    template <bool WANT_NODE>
    __device__ __forceinline__ int pick(const OtfArgs& a, int32_t prev, int64_t base, int K, int kk, double u2, int lane,
                                        bool keep_row) {
        int32_t* P = rows2 + pb * kOtfRow;
        int32_t* C = rows2 + (pb ^ 1) * kOtfRow;
        // has_edge(nbr, prev) searches prev's row in LDS.  It was `cur` one step ago, so the row staged then is reused; it
        // is fetched only after a stored-table step, for another walk, or when it did not fit.
        if (prev >= 0 && a.g.symmetric) {
            if (cached_node != prev) { cached_n = n2v::wave_cache_row(a.g, P, prev, lane, kOtfRow); cached_node = prev; }
            ws.row = P;
            ws.row_n = cached_n;
        } else {
            ws.row_n = -1;
        }
        const int32_t nb0 = lane < K ? a.g.col[base + lane] : -1;   // the row's first 64 entries, one per lane
        const bool staged = keep_row && a.g.symmetric && K <= kOtfRow;
        if (staged) {
            if (lane < K) C[lane] = nb0;
            for (int i = 64 + lane; i < K; i += 64) C[i] = a.g.col[base + i];
            n2v::wave_sync();
        }
        int pk = kk;
        if (K <= 64 && a.draw_first) {                                // the table in registers, one slot per lane
            pk = n2v::wave_draw_le64(a.g, ws, prev, base, nb0, K, kk, u2, a.draw_first == 2 && a.exact_sum, a.wp, a.wq, lane);
#ifdef N2V_OTF_LAB_ALWAYS_ACCEPT   /* timing ceiling of the fast path only: WRONG walks */
            pk = pk < 0 ? pk : kk;
#endif
            if (pk < 0) return -1;
        } else {
            bool drawn = false;
            if (a.draw_first == 2) {         // dyadic weights: the pick from counts and a sweep over the three weight classes
                const int d = n2v::dyadic_draw(a.g, ws, prev, base, K, kk, u2, a.wp, a.wq, a.exact_sum != 0,
                                               reinterpret_cast<int32_t*>(Tl), kLdsSlots * 4, reinterpret_cast<int32_t*>(ws.feed), lane);
                if (d >= 0) { pk = d; drawn = true; }   // -2: more common neighbours than the sweep's list holds -> build
            }
#ifdef N2V_OTF_LAB_ALWAYS_ACCEPT
            drawn = true;
#endif
            if (!drawn) {
                n2v_alias_slot* T = K <= kLdsSlots ? Tl : Tg;
                double norm;
                if (!n2v::wave_weights_and_norm(a.g, T, ws, prev, base, K, lane, norm)) return -1;
                if (a.draw_first == 1) {     // slot kk `smaller` (its q is final, :253-255) and accepted (:278)?
                    const double q0 = (double)K * (T[kk].q / norm);
                    drawn = q0 < 1.0 && u2 < q0;
                }
                if (!drawn) {
                    n2v::wave_finish_table(T, K, norm, lane);
                    pk = (u2 < T[kk].q) ? kk : T[kk].J;  // :278-281
                }
            }
        }
        pk = uni(pk);
        if (WANT_NODE)
            picked_node = K <= 64 ? __builtin_amdgcn_readlane(nb0, pk) : staged ? uni(C[pk]) : uni(a.g.col[base + pk]);
        if (keep_row) {
            pb ^= 1;                          // the staged row is the next call's prev row
            cached_node = -2;                 // the caller names it (set_cached) once it knows the node id
            cached_n = staged ? K : -1;
        }
        __builtin_amdgcn_wave_barrier();      // the table is rebuilt in place by the next call
        return pk;
    }
    __device__ __forceinline__ void set_cached(int32_t node) { if (cached_node == -2) cached_node = cached_n >= 0 ? node : -1; }
};

#define N2V_OTF_WAVE_STATE(W)                                                                                      \
    __shared__ n2v_alias_slot lds[4 * kLdsSlots];                                                                  \
    __shared__ double feed[4 * n2v::kFeed];                                                                        \
    __shared__ int32_t rows[4 * 2 * kOtfRow];                                                                      \
    const int lane = threadIdx.x & 63;                                                                             \
    /* the wave index is the same in all 64 lanes: tell the compiler, so that everything derived from it (walk id, \
       loop bounds, table sizes) is scalar and loops branch on SCC instead of EXEC */                              \
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                               \
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + wv;                                                      \
    const int64_t n_waves = (int64_t)gridDim.x * 4;                                                                \
    OtfWave W{lds + wv * kLdsSlots, a.scratch + wave_global * a.max_degree,                                        \
              n2v::WaveScratch{feed + wv * n2v::kFeed, rows + wv * 2 * kOtfRow, -1}, rows + wv * 2 * kOtfRow, 0, -1, -1, -1}

// ---- one wavefront per walk: every step without a stored table (all of them when HYBRID is false) ------------------
// HYBRID: small launches of the budgeted walk (fewer walks than lanes on the chip) — one walk per wave keeps more rebuilds
// in flight than the lane-per-walk kernel below, which serves its 64 lanes' rebuilds one after the other.
template <bool HYBRID>
__global__ void __launch_bounds__(256) walk_otf_kernel(OtfArgs a) {
    N2V_OTF_WAVE_STATE(W);
    const int32_t L = a.L;

    for (int64_t lw = wave_global; lw < a.n_local; lw += n_waves) {
        const int64_t rl = lw / a.pos_count, pl = lw - rl * a.pos_count;
        const uint64_t gw = (uint64_t)((a.round_begin + rl) * a.n_starts + a.pos_begin + pl);
        int32_t cur = uni(a.starts[a.pos_begin + pl]), prev = -1;
        int32_t* out = a.walks + lw * (int64_t)L;
        const double* up = nullptr;
        if (a.rng_mode == N2V_RNG_UNIFORMS)
            up = a.uniforms + (a.walk_uoff ? a.walk_uoff[lw] : (int64_t)2 * (L - 1) * lw);
        if (lane == 0) out[0] = cur;
        int32_t len = 1;
        bool failed = false;
        // hybrid: the stored table of the step about to be taken (first step: the node table), or N2V_NO_TABLE
        const n2v_fat_slot* arr = a.node_fat;
        uint64_t tbl = HYBRID ? (uint64_t)uni64(a.g.row_ptr[cur]) : (uint64_t)N2V_NO_TABLE;
        int K = HYBRID ? uni((int)(a.g.row_ptr[cur + 1] - a.g.row_ptr[cur])) : 0;
        for (; len < L; ++len) {
            int64_t base = 0;
            if (!HYBRID || tbl == (uint64_t)N2V_NO_TABLE) {
                base = uni64(a.g.row_ptr[cur]);
                K = uni((int)(a.g.row_ptr[cur + 1] - base));
            }
            if (K == 0) break;  // dead end (:50-51)
            double u1, u2;
            const uint32_t t = (uint32_t)(len - 1);
            if (a.rng_mode == N2V_RNG_UNIFORMS) { u1 = up[2 * (int64_t)t]; u2 = up[2 * (int64_t)t + 1]; }
            else n2v::philox_uniforms(a.seed, gw, t, u1, u2);
            const int kk = (int)(u1 * (double)K);  // :277
            if (HYBRID && tbl != (uint64_t)N2V_NO_TABLE) {
                // stored table: one 32-B slot (every lane reads the same address), both outcomes' records inside
                const uint4* sp = reinterpret_cast<const uint4*>(arr + tbl + (uint64_t)kk);
                const uint4 lo = sp[0], hi = sp[1];
                const double qk = __hiloint2double((int)lo.y, (int)lo.x);
                const bool keep = u2 < qk;  // :278
                const uint32_t slot_lo = keep ? lo.z : hi.y, deg_hi = keep ? lo.w : hi.z, dst = keep ? hi.x : hi.w;
                prev = cur;
                cur = uni((int32_t)dst);
                tbl = ((uint64_t)(uint32_t)uni((int)(deg_hi >> 24)) << 32) | (uint32_t)uni((int)slot_lo);
                K = uni((int)(deg_hi & 0xFFFFFFu));
                arr = a.fat;
            } else {
                const int pick = W.pick<!HYBRID>(a, prev, base, K, kk, u2, lane, true);
                if (pick < 0) { failed = true; break; }
                prev = cur;
                W.set_cached(prev);
                if (HYBRID) {
                    const uint4 r = *reinterpret_cast<const uint4*>(a.recs + base + pick);   // {slot_lo, base, dst, deg_hi}
                    cur = uni((int32_t)r.z);
                    tbl = ((uint64_t)(uint32_t)uni((int)(r.w >> 24)) << 32) | (uint32_t)uni((int)r.x);
                    K = uni((int)(r.w & 0xFFFFFFu));
                    arr = a.fat;
                } else {
                    cur = W.picked_node;
                }
            }
            if (lane == 0) out[len] = cur;
        }
        if (failed && lane == 0) atomicOr(a.status, N2V_STATUS_ZERO_NORM);
        if (lane == 0) {
            a.lens[lw] = len;
            for (int32_t i = len; i < L; ++i) out[i] = -1;
        }
    }
}

// ---- one LANE per walk (tables under a memory budget) -------------------------------------------------------------------
// The table-driven walk's layout (n2v_walk_fat.hip): each lane follows its own walk through the stored fat slots.  A lane
// whose next table is not stored (record index N2V_NO_TABLE) raises its bit; the wave then serves those lanes one after
// the other with the whole-wave step above (its arguments broadcast by v_readlane) and hands each its record.  With most
// steps stored this keeps 64 walks per wave in flight instead of one.
template <int RNG, int BURST>
__global__ void __launch_bounds__(256) walk_hybrid_lanes_kernel(OtfArgs a) {
    N2V_OTF_WAVE_STATE(W);
    const int32_t L = a.L;
    for (int64_t lw0 = wave_global * 64; lw0 < a.n_local; lw0 += n_waves * 64) {
        const int64_t lw = lw0 + lane;
        const bool mine = lw < a.n_local;
        int64_t rl = 0, pl = 0;
        if (mine) { rl = lw / a.pos_count; pl = lw - rl * a.pos_count; }
        const uint64_t gw = (uint64_t)((a.round_begin + rl) * a.n_starts + a.pos_begin + pl);
        int32_t cur = mine ? a.starts[a.pos_begin + pl] : 0, prev = -1;
        int64_t b0 = 0, b1 = 0;
        if (mine) { b0 = a.g.row_ptr[cur]; b1 = a.g.row_ptr[cur + 1]; }
        const n2v_fat_slot* arr = a.node_fat;          // first step: the node table (:69-70), always stored
        uint64_t tbl = (uint64_t)b0;
        uint32_t K = (uint32_t)(b1 - b0);
        int32_t len = 1;
        uint32_t t = 0;
        bool failed = false;
        const double* up = nullptr;
        if (RNG == N2V_RNG_UNIFORMS && mine) up = a.uniforms + (a.walk_uoff ? a.walk_uoff[lw] : (int64_t)2 * (L - 1) * lw);

        auto step = [&]() -> int32_t {
            const bool live = mine && K != 0 && !failed;   // dead end: stop, consume nothing (:76-77)
            double u1 = 0.0, u2 = 0.0;
            if (live) {
                if (RNG == N2V_RNG_UNIFORMS) { u1 = up[2 * (int64_t)t]; u2 = up[2 * (int64_t)t + 1]; }
                else n2v::philox_uniforms(a.seed, gw, t, u1, u2);
                ++t;
            }
            const uint32_t kk = (uint32_t)(u1 * (double)K);  // :277
            const bool stored = live && tbl != (uint64_t)N2V_NO_TABLE;
            uint32_t slot_lo = 0, deg_hi = 0, dst = 0;
            if (stored) {
                const uint4* sp = reinterpret_cast<const uint4*>(arr + tbl + (uint64_t)kk);
                const uint4 lo = sp[0], hi = sp[1];
                const bool keep = u2 < __hiloint2double((int)lo.y, (int)lo.x);  // :278
                slot_lo = keep ? lo.z : hi.y; deg_hi = keep ? lo.w : hi.z; dst = keep ? hi.x : hi.w;
            }
            unsigned long long need = __ballot(live && !stored);
            while (need != 0ULL) {
                const int j = __builtin_ctzll(need);
                need &= need - 1ULL;
                const int32_t s_prev = __builtin_amdgcn_readlane(prev, j), s_cur = __builtin_amdgcn_readlane(cur, j);
                const int s_K = __builtin_amdgcn_readlane((int)K, j), s_kk = __builtin_amdgcn_readlane((int)kk, j);
                const double s_u2 = n2v::readlane_f64(u2, j);
                const int64_t base = uni64(a.g.row_ptr[s_cur]);
                const int pk = W.pick<false>(a, s_prev, base, s_K, s_kk, s_u2, lane, false);
                if (pk < 0) { if (lane == j) failed = true; continue; }
                const uint4 r = *reinterpret_cast<const uint4*>(a.recs + base + pk);   // {slot_lo, base, dst, deg_hi}
                if (lane == j) { slot_lo = r.x; deg_hi = r.w; dst = r.z; }
            }
            if (!live || failed) return -1;
            prev = cur;
            cur = (int32_t)dst;
            tbl = ((uint64_t)(deg_hi >> 24) << 32) | slot_lo;
            K = deg_hi & 0xFFFFFFu;
            arr = a.fat;
            ++len;
            return cur;
        };

        int32_t* out = a.walks + lw * (int64_t)L;
        int32_t buf[BURST];
        buf[0] = cur;
#pragma unroll
        for (int i = 1; i < BURST; ++i) buf[i] = step();
        for (int32_t g = 0;;) {
            if (mine) {
                if (BURST >= 4) {
#pragma unroll
                    for (int i = 0; i + 3 < BURST; i += 4) {
                        typedef int v4i __attribute__((ext_vector_type(4)));
                        v4i v = {buf[i], buf[i + 1], buf[i + 2], buf[i + 3]};
                        *reinterpret_cast<v4i*>(out + g + i) = v;
                    }
                } else {
                    out[g] = buf[0];
                }
            }
            g += BURST;
            if (g >= L) break;
#pragma unroll
            for (int i = 0; i < BURST; ++i) buf[i] = step();
        }
        if (mine) a.lens[lw] = len;
        if (__ballot(failed) != 0ULL && lane == 0) atomicOr(a.status, N2V_STATUS_ZERO_NORM);
    }
}

}  // namespace

namespace {
int launch_otf(const char* who, bool hybrid, const int64_t* row_ptr, const int32_t* col, const double* w, double p, double q,
               int32_t symmetric, int64_t max_degree, const n2v_fat_slot* node_fat, const n2v_fat_slot* fat,
               const n2v_edge_rec* recs, const int32_t* starts, int64_t n_starts, int64_t pos_begin,
               int64_t pos_count, int64_t round_begin, int64_t round_count, int32_t walk_length,
               int32_t rng_mode, const double* uniforms, const int64_t* walk_uoff, uint64_t seed,
               n2v_alias_slot* scratch, int64_t scratch_slots, int32_t* walks, int32_t* lens,
               int32_t* status, void* stream) {
    if (pos_count < 0 || round_count < 0 || pos_begin < 0 || round_begin < 0 || walk_length < 1 ||
        pos_begin + pos_count > n_starts || max_degree < 0)
        return n2v::fail(N2V_ERR_INVALID, "%s: bad shard or length", who);
    const int64_t n_local = pos_count * round_count;
    if (n_local == 0) return N2V_OK;
    if (!row_ptr || !col || !starts || !walks || !lens || !status)
        return n2v::fail(N2V_ERR_INVALID, "%s: null pointer", who);
    if (hybrid && (!node_fat || !recs || (walk_length > 2 && !fat)))
        return n2v::fail(N2V_ERR_INVALID, "%s: the stored tables and the walk records are needed", who);
    if (hybrid && ((((uintptr_t)node_fat | (uintptr_t)fat) & 31) != 0))
        return n2v::fail(N2V_ERR_INVALID, "%s: fat slots not 32-byte aligned", who);
    if (!(p == p) || !(q == q) || p == 0.0 || q == 0.0)
        return n2v::fail(N2V_ERR_INVALID, "%s: p and q must be non-zero numbers", who);
    if (rng_mode != N2V_RNG_UNIFORMS && rng_mode != N2V_RNG_PHILOX)
        return n2v::fail(N2V_ERR_INVALID, "%s: rng_mode %d", who, (int)rng_mode);
    if (rng_mode == N2V_RNG_UNIFORMS && walk_length > 1 && !uniforms)
        return n2v::fail(N2V_ERR_INVALID, "%s: parity mode needs a uniform buffer", who);
    // grid: as many resident waves as the scratch rows allow (4 workgroups of 4 waves per CU by LDS)
    int64_t blocks = (n_local + 3) / 4;
    if (blocks > 256 * kWgPerCu) blocks = 256 * kWgPerCu;   // every resident workgroup slot (LDS-bound), once
#ifdef N2V_OTF_LAB_GRID_DIV
    blocks = (blocks + N2V_OTF_LAB_GRID_DIV - 1) / N2V_OTF_LAB_GRID_DIV;
#endif
    if (max_degree > kLdsSlots) {
        if (!scratch) return n2v::fail(N2V_ERR_INVALID, "%s: scratch needed (max degree %lld > %d)", who,
                                       (long long)max_degree, kLdsSlots);
        const int64_t fit = scratch_slots / max_degree / 4;
        if (fit < 1) return n2v::fail(N2V_ERR_INVALID, "%s: scratch smaller than 4 x max_degree slots", who);
        if (blocks > fit) blocks = fit;
    }
    // three weight classes (n2v_wave_table.h): unweighted, undirected — 2; their sum a count: 1/p and 1/q multiples of
    // 2^-20 up to 2^10, degrees < 2^21
    const double wp = 1.0 / p, wq = 1.0 / q;
    auto dyadic = [](double x) { return x > 0.0 && x <= 1024.0 && x * 1048576.0 == (double)(int64_t)(x * 1048576.0); };
    int32_t draw_first = (!w && symmetric) ? 2 : 1;
    const int32_t exact_sum = dyadic(wp) && dyadic(wq) && max_degree < (1 << 21);
#ifdef N2V_OTF_LAB_DRAW_FIRST   /* tools/lab/otf_variants.sh: 0 = always build the table, 1 = never count */
    draw_first = N2V_OTF_LAB_DRAW_FIRST < draw_first ? N2V_OTF_LAB_DRAW_FIRST : draw_first;
#endif
    OtfArgs a{n2v::RowCtx{row_ptr, col, w, p, q, symmetric}, starts, n_starts, pos_begin, pos_count, round_begin, n_local, walk_length,
              rng_mode, uniforms, walk_uoff, seed, scratch, max_degree, node_fat, fat, recs, walks, lens, status, draw_first, exact_sum, wp, wq};
    hipStream_t st = (hipStream_t)stream;
    // budgeted walk: one lane per walk once the launch is large enough to fill half the lanes of the resident waves
    if (hybrid && n_local >= blocks * 128) {
        int64_t lblocks = (n_local + 255) / 256;
        if (lblocks > blocks) lblocks = blocks;
        const dim3 grid((unsigned)lblocks), block(256);
        // 16-B output pieces: the 64-B bursts of the table-driven kernel cost 30 VGPRs here (4 instead of 6 waves per SIMD)
        // and this kernel lives on the waves it keeps in flight (C3, a third of the tables: 5.6 -> 6.3e9 steps/s)
        const int burst = (walk_length % 4 == 0 && ((uintptr_t)walks & 15) == 0) ? 4 : 1;
#define N2V_LAUNCH_LANES(RNG)                                                                                       \
        do {                                                                                                        \
            if (burst == 4) hipLaunchKernelGGL((walk_hybrid_lanes_kernel<RNG, 4>), grid, block, 0, st, a);          \
            else hipLaunchKernelGGL((walk_hybrid_lanes_kernel<RNG, 1>), grid, block, 0, st, a);                     \
        } while (0)
        if (rng_mode == N2V_RNG_UNIFORMS) N2V_LAUNCH_LANES(N2V_RNG_UNIFORMS);
        else N2V_LAUNCH_LANES(N2V_RNG_PHILOX);
#undef N2V_LAUNCH_LANES
    } else if (hybrid) {
        hipLaunchKernelGGL(walk_otf_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(walk_otf_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    }
    return n2v::check_launch(who);
}
}  // namespace

extern "C" int32_t n2v_walk_otf_lds_slots(void) { return kLdsSlots; }
extern "C" int32_t n2v_walk_otf_max_waves(void) { return 256 * kWgPerCu * 4; }

extern "C" int n2v_walk_on_the_fly(const int64_t* row_ptr, const int32_t* col, const double* w, double p, double q,
                                   int32_t symmetric, int64_t max_degree, const int32_t* starts, int64_t n_starts, int64_t pos_begin,
                                   int64_t pos_count, int64_t round_begin, int64_t round_count, int32_t walk_length,
                                   int32_t rng_mode, const double* uniforms, const int64_t* walk_uoff, uint64_t seed,
                                   n2v_alias_slot* scratch, int64_t scratch_slots, int32_t* walks, int32_t* lens,
                                   int32_t* status, void* stream) {
    return launch_otf("n2v_walk_on_the_fly", false, row_ptr, col, w, p, q, symmetric, max_degree, nullptr, nullptr, nullptr, starts,
                      n_starts, pos_begin, pos_count, round_begin, round_count, walk_length, rng_mode, uniforms, walk_uoff,
                      seed, scratch, scratch_slots, walks, lens, status, stream);
}

extern "C" int n2v_walk_hybrid(const int64_t* row_ptr, const int32_t* col, const double* w, double p, double q, int32_t symmetric,
                               int64_t max_degree, const n2v_fat_slot* node_fat, const n2v_fat_slot* fat,
                               const n2v_edge_rec* recs, const int32_t* starts, int64_t n_starts, int64_t pos_begin,
                               int64_t pos_count, int64_t round_begin, int64_t round_count, int32_t walk_length,
                               int32_t rng_mode, const double* uniforms, const int64_t* walk_uoff, uint64_t seed,
                               n2v_alias_slot* scratch, int64_t scratch_slots, int32_t* walks, int32_t* lens,
                               int32_t* status, void* stream) {
    return launch_otf("n2v_walk_hybrid", true, row_ptr, col, w, p, q, symmetric, max_degree, node_fat, fat, recs, starts,
                      n_starts, pos_begin, pos_count, round_begin, round_count, walk_length, rng_mode, uniforms, walk_uoff,
                      seed, scratch, scratch_slots, walks, lens, status, stream);
}
