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
// The table lives in the wave's slice of LDS while K <= kLdsSlots and in a per-wave global
// scratch row of max_degree slots otherwise.  Serial chains of many waves interleave on a
// SIMD, so throughput comes from occupancy; -ffp-contract=off keeps every rounding separate.
#include "n2v_common.h"
#include "n2v_wave_table.h"

namespace {

constexpr int kLdsSlots = 512;  // 8 KiB of LDS per wave, 32 KiB per 4-wave workgroup

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
    int32_t* walks;
    int32_t* lens;
    int32_t* status;
};

__global__ void __launch_bounds__(256) walk_otf_kernel(OtfArgs a) {
    __shared__ n2v_alias_slot lds[4 * kLdsSlots];
    __shared__ double feed[4 * n2v::kFeed];
    __shared__ int32_t rows[4 * n2v::kRowCache];
    const int lane = threadIdx.x & 63;
    // the wave index is the same in all 64 lanes: tell the compiler, so that everything derived from
    // it (walk id, loop bounds, table sizes) is scalar and loops branch on SCC instead of EXEC
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    n2v_alias_slot* Tl = lds + wv * kLdsSlots;
    int32_t* my_row = rows + wv * n2v::kRowCache;
    n2v::WaveScratch ws{feed + wv * n2v::kFeed, my_row, -1};
    const int64_t wave_global = (int64_t)blockIdx.x * 4 + wv;
    n2v_alias_slot* Tg = a.scratch + wave_global * a.max_degree;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
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
        for (; len < L; ++len) {
            const int64_t base = uni64(a.g.row_ptr[cur]);
            const int K = uni((int)(a.g.row_ptr[cur + 1] - base));
            if (K == 0) break;  // dead end (:50-51)
            bool ok;
            ws.row_n = n2v::wave_cache_row(a.g, my_row, prev, lane);     // has_edge(nbr, prev): prev's row, staged in LDS
            if (K <= kLdsSlots) ok = n2v::wave_build_table(a.g, Tl, ws, prev, base, K, lane);
            else ok = n2v::wave_build_table(a.g, Tg, ws, prev, base, K, lane);
            if (!ok) { failed = true; break; }
            double u1, u2;
            const uint32_t t = (uint32_t)(len - 1);
            if (a.rng_mode == N2V_RNG_UNIFORMS) { u1 = up[2 * (int64_t)t]; u2 = up[2 * (int64_t)t + 1]; }
            else n2v::philox_uniforms(a.seed, gw, t, u1, u2);
            const int kk = (int)(u1 * (double)K);  // :277
            double qk; int Jk;
            if (K <= kLdsSlots) { qk = Tl[kk].q; Jk = Tl[kk].J; }
            else { qk = Tg[kk].q; Jk = Tg[kk].J; }
            const int pick = (u2 < qk) ? kk : Jk;  // :278-281
            prev = cur;
            cur = uni(a.g.col[base + pick]);
            if (lane == 0) out[len] = cur;
            __builtin_amdgcn_wave_barrier();  // the table is rebuilt in place on the next step
        }
        if (failed && lane == 0) atomicOr(a.status, N2V_STATUS_ZERO_NORM);
        if (lane == 0) {
            a.lens[lw] = len;
            for (int32_t i = len; i < L; ++i) out[i] = -1;
        }
    }
}

}  // namespace

extern "C" int n2v_walk_on_the_fly(const int64_t* row_ptr, const int32_t* col, const double* w, double p, double q,
                                   int32_t symmetric, int64_t max_degree, const int32_t* starts, int64_t n_starts, int64_t pos_begin,
                                   int64_t pos_count, int64_t round_begin, int64_t round_count, int32_t walk_length,
                                   int32_t rng_mode, const double* uniforms, const int64_t* walk_uoff, uint64_t seed,
                                   n2v_alias_slot* scratch, int64_t scratch_slots, int32_t* walks, int32_t* lens,
                                   int32_t* status, void* stream) {
    if (pos_count < 0 || round_count < 0 || pos_begin < 0 || round_begin < 0 || walk_length < 1 ||
        pos_begin + pos_count > n_starts || max_degree < 0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk_on_the_fly: bad shard or length");
    const int64_t n_local = pos_count * round_count;
    if (n_local == 0) return N2V_OK;
    if (!row_ptr || !col || !starts || !walks || !lens || !status)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk_on_the_fly: null pointer");
    if (!(p == p) || !(q == q) || p == 0.0 || q == 0.0)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk_on_the_fly: p and q must be non-zero numbers");
    if (rng_mode != N2V_RNG_UNIFORMS && rng_mode != N2V_RNG_PHILOX)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk_on_the_fly: rng_mode %d", (int)rng_mode);
    if (rng_mode == N2V_RNG_UNIFORMS && walk_length > 1 && !uniforms)
        return n2v::fail(N2V_ERR_INVALID, "n2v_walk_on_the_fly: parity mode needs a uniform buffer");
    // grid: as many resident waves as the scratch rows allow (4 workgroups of 4 waves per CU by LDS)
    int64_t blocks = (n_local + 3) / 4;
    if (blocks > 256 * 4) blocks = 256 * 4;   // 40 KiB of LDS per workgroup: 4 per CU
    if (max_degree > kLdsSlots) {
        if (!scratch) return n2v::fail(N2V_ERR_INVALID, "n2v_walk_on_the_fly: scratch needed (max degree %lld > %d)",
                                       (long long)max_degree, kLdsSlots);
        const int64_t fit = scratch_slots / max_degree / 4;
        if (fit < 1) return n2v::fail(N2V_ERR_INVALID, "n2v_walk_on_the_fly: scratch smaller than 4 x max_degree slots");
        if (blocks > fit) blocks = fit;
    }
    OtfArgs a{n2v::RowCtx{row_ptr, col, w, p, q, symmetric}, starts, n_starts, pos_begin, pos_count, round_begin, n_local, walk_length,
              rng_mode, uniforms, walk_uoff, seed, scratch, max_degree, walks, lens, status};
    hipLaunchKernelGGL(walk_otf_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return n2v::check_launch("n2v_walk_on_the_fly");
}
