// Vose's alias pairing exactly as the reference runs it (src/node2vec.py:246-269), written so that
// the serial loop carries its dependency through registers instead of through memory.
//
// Reference semantics: `smaller` and `larger` are Python lists filled in index order and popped
// from the end; each iteration pops one index from both, sets J[small] = large,
// q[large] = q[large] + q[small] - 1.0 (left to right) and pushes `large` back onto one of them.
//
// Observation: every push is followed by a pop of BOTH stacks, so at most one pushed element is
// ever pending, and it is always the next one popped from its stack.  It can therefore live in a
// register (index + current q) and never touch memory until it is final.  The memory stacks (kept
// in the `aux` word of the table's own slots: `smaller` grows up from slot 0, `larger` grows down
// from slot K-1) then hold only the entries of the initial classification, which are popped in a
// fixed order and whose q is never modified before they are popped — so they can be prefetched a
// few entries ahead.  What remains on the critical path of an iteration is two fp64 adds and a
// compare; the loads are independent streams.  Same result bit for bit (checked against the golden
// vectors), ~an order of magnitude less latency per slot for long tables (hubs).
#pragma once
#include "n2v_hip.h"

namespace n2v {

// What T[k].q holds on entry: kScaled = K*prob already; kProb = the normalised probability, multiplied
// by K here (:253); kWeight = the unnormalised weight u, turned into K * (u / norm) here (:150 then
// :253, two roundings as in the reference) so that the caller needs no separate normalising pass.
// SlotPtr: n2v_alias_slot* into global memory or LDS.
enum : int { kScaled = 0, kProb = 1, kWeight = 2 };

template <int INPUT, typename SlotPtr>
__device__ __forceinline__ void vose_pair(SlotPtr T, int64_t K, double norm = 1.0) {
    int64_t ns = 0, nl = 0;
    const double Kd = (double)K;
    for (int64_t k = 0; k < K; ++k) {  // :252-257
        double qk = T[k].q;
        if (INPUT != kScaled) {
            if (INPUT == kWeight) qk = qk / norm;
            qk = Kd * qk;
            T[k].q = qk;
        }
        T[k].J = 0;
        if (qk < 1.0) T[ns++].aux = (int32_t)k;
        else T[K - (++nl)].aux = (int32_t)k;
    }
    if (ns == 0 || nl == 0) return;

    // lookahead over the memory `smaller` stack (pop order: positions ns-1, ns-2, ...)
    int64_t next_pos = ns - 1;  // position of the next entry to PREFETCH
    int32_t p0i = 0, p1i = 0, p2i = 0, p3i = 0;
    double p0q = 0, p1q = 0, p2q = 0, p3q = 0;
    auto fetch = [&](int32_t& pi, double& pq) {
        if (next_pos >= 0) {
            pi = T[next_pos].aux;
            pq = T[pi].q;
            --next_pos;
        }
    };
    fetch(p0i, p0q);
    fetch(p1i, p1q);
    fetch(p2i, p2q);
    fetch(p3i, p3q);

    int64_t mem_s = ns, mem_l = nl;  // entries left in the memory stacks
    bool hasS = false, hasL = false;  // a pushed element pending in registers (top of its stack)
    int32_t rsi = 0, rli = 0;
    double rsq = 0, rlq = 0;
    while ((mem_s > 0 || hasS) && (mem_l > 0 || hasL)) {  // :259
        int32_t small, large;
        double qs, ql;
        if (hasS) {  // smaller.pop(): the element pushed by the previous iteration
            small = rsi;
            qs = rsq;
            hasS = false;
            T[small].q = qs;  // final value of that slot
        } else {
            small = p0i;
            qs = p0q;
            p0i = p1i; p0q = p1q;
            p1i = p2i; p1q = p2q;
            p2i = p3i; p2q = p3q;
            fetch(p3i, p3q);
            --mem_s;
        }
        if (hasL) {  // larger.pop()
            large = rli;
            ql = rlq;
            hasL = false;
        } else {
            large = T[K - mem_l].aux;
            --mem_l;
            ql = T[large].q;
        }
        T[small].J = large;      // :263
        double t = ql + qs;      // :264, left to right
        t = t - 1.0;
        if (t < 1.0) {           // :265-268
            hasS = true; rsi = large; rsq = t;
        } else {
            hasL = true; rli = large; rlq = t;
        }
    }
    if (hasS) T[rsi].q = rsq;
    if (hasL) T[rli].q = rlq;
}

}  // namespace n2v
