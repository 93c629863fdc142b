// One alias table built by the 64 lanes of one wavefront — shared by the edge-table builder (n2v_tables.hip) and the
// on-the-fly walk (n2v_walk_otf.hip).  Same bits as alias_setup / get_alias_edge of the reference
// (src/node2vec.py:133-152, 240-269): the neighbour classification (:142-148), the normalisation (:150, :253) and the
// initial fill of Vose's two stacks (:252-257) run in parallel with coalesced row reads; the two inherently serial
// pieces keep the reference's order — the left-to-right fp64 sum (:149) and the stack pairing (:259-268) — but are
// fed from registers: 64 entries of a stack are loaded by the 64 lanes at once and consumed through v_readlane, and
// an element pushed back is always the next one popped from its stack, so it never touches memory (n2v_vose.h has
// the argument).  T may live in LDS or in global memory.  Compile with -ffp-contract=off.
#pragma once
#include "n2v_common.h"

namespace n2v {

__device__ __forceinline__ void wave_sync() {  // order this wave's LDS / global traffic between phases
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// wave-uniform values that arrive through vector registers: make them scalar for the compiler, so the serial
// loops branch on SCC and keep their counters in SGPRs
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int64_t uni64(int64_t v) {
    return ((int64_t)uni((int)(v >> 32)) << 32) | (uint32_t)uni((int)v);
}
__device__ __forceinline__ double unid(double v) {
    return __hiloint2double(uni(__double2hiint(v)), uni(__double2loint(v)));
}
__device__ __forceinline__ double readlane_f64(double v, int j) {  // j wave-uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
    return __hiloint2double(hi, lo);
}

// What a table is built from: the sorted CSR and the walk's p, q.
struct RowCtx {
    const int64_t* row_ptr;
    const int32_t* col;
    const double* w;      // NULL: all weights 1
    double p, q;
    int32_t symmetric;    // undirected graph: has_edge(nbr, src) == nbr in row(src), one shared row
};

// Builds the alias table of the step that arrives at the node whose row starts at `base` (K neighbours) from `src`
// (src < 0: the first step, i.e. the node table of src/node2vec.py:184-188) in T[0..K): afterwards T[k].q / T[k].J
// are q[k] / J[k] of alias_setup.  All 64 lanes of the wave call it together.  Returns false when the weights sum
// to 0 (the reference raises ZeroDivisionError, :150 / :187).
template <typename Slot>
__device__ __forceinline__ bool wave_build_table(const RowCtx& a, Slot* T, int32_t src, int64_t base, int K, int lane) {
    // ---- 1. unnormalised weights in parallel (:142-148); has_edge(nbr, src) is "nbr in row(src)" on an undirected
    //         graph, so all lanes probe ONE row
    for (int k = lane; k < K; k += 64) {
        const int32_t nb = a.col[base + k];
        const double wt = a.w ? a.w[base + k] : 1.0;
        double u;
        if (src < 0) u = wt;
        else if (nb == src) u = wt / a.p;
        else if (a.symmetric ? row_contains(a.row_ptr, a.col, src, nb) : row_contains(a.row_ptr, a.col, nb, src)) u = wt;
        else u = wt / a.q;
        T[k].q = u;
    }
    wave_sync();
    // ---- 2. norm = sum(unnormalized_probs), strictly left to right (:149): 64 values per coalesced load, consumed
    //         in order through v_readlane by every lane alike
    double norm = 0.0;
    for (int c = 0; c < K; c += 64) {
        const double v = (c + lane < K) ? T[c + lane].q : 0.0;
        const int cnt = min(64, K - c);
        for (int j = 0; j < cnt; ++j) norm = norm + readlane_f64(v, j);
    }
    norm = unid(norm);
    if (norm == 0.0) return false;
    // ---- 3. q = K * (u / norm) (:150 then :253, two roundings) and the two index stacks in index order
    //         (:252-257): `smaller` grows up from slot 0, `larger` down from slot K-1
    const double Kd = (double)K;
    int ns = 0, nl = 0;
    for (int c = 0; c < K; c += 64) {
        const int k = c + lane;
        const bool valid = k < K;
        double qk = 0.0;
        if (valid) {
            qk = Kd * (T[k].q / norm);
            T[k].q = qk;
            T[k].J = 0;
        }
        const bool is_small = valid && (qk < 1.0);
        const unsigned long long ms = __ballot(is_small), ml = __ballot(valid && !is_small);
        const unsigned long long below = (1ULL << lane) - 1ULL;
        if (is_small) T[ns + __popcll(ms & below)].aux = k;
        else if (valid) T[K - (nl + __popcll(ml & below) + 1)].aux = k;
        ns += __popcll(ms);
        nl += __popcll(ml);
    }
    ns = uni(ns);
    nl = uni(nl);
    wave_sync();
    // ---- 4. pairing (:259-268).  Both memory stacks only ever hold entries of the initial classification, popped
    //         in a fixed order and never modified before they are popped: they are streamed 64 entries at a time,
    //         one per lane, and handed to the (wave-uniform) loop by v_readlane.
    int mem_s = ns, mem_l = nl;
    bool hasS = false, hasL = false;
    int rsi = 0, rli = 0;
    double rsq = 0.0, rlq = 0.0;
    int si = 0, s_cnt = 0, s_pos = 0, li = 0, l_cnt = 0, l_pos = 0;
    double sq = 0.0, lq = 0.0;
    while ((mem_s > 0 || hasS) && (mem_l > 0 || hasL)) {
        int small, large;
        double qs, ql;
        if (hasS) {                       // smaller.pop(): the element the previous iteration pushed
            small = rsi; qs = rsq; hasS = false;
            if (lane == 0) T[small].q = qs;
        } else {
            if (s_pos == s_cnt) {         // next <= 64 entries of `smaller`, in pop order (top = position mem_s-1)
                const int pos = mem_s - 1 - lane;
                if (pos >= 0) { si = T[pos].aux; sq = T[si].q; }
                s_cnt = min(64, mem_s);
                s_pos = 0;
            }
            small = __builtin_amdgcn_readlane(si, s_pos);
            qs = readlane_f64(sq, s_pos);
            ++s_pos;
            --mem_s;
        }
        if (hasL) {                       // larger.pop()
            large = rli; ql = rlq; hasL = false;
        } else {
            if (l_pos == l_cnt) {         // next <= 64 entries of `larger` (top = position K-mem_l, then upwards)
                const int pos = K - mem_l + lane;
                if (pos < K) { li = T[pos].aux; lq = T[li].q; }
                l_cnt = min(64, mem_l);
                l_pos = 0;
            }
            large = __builtin_amdgcn_readlane(li, l_pos);
            ql = readlane_f64(lq, l_pos);
            ++l_pos;
            --mem_l;
        }
        if (lane == 0) T[small].J = large;          // :263
        double t = ql + qs;                          // :264, left to right
        t = t - 1.0;
        if (uni((int)(t < 1.0))) { hasS = true; rsi = large; rsq = t; }
        else { hasL = true; rli = large; rlq = t; }
    }
    if (lane == 0) {
        if (hasS) T[rsi].q = rsq;
        if (hasL) T[rli].q = rlq;
    }
    wave_sync();
    return true;
}

}  // namespace n2v
