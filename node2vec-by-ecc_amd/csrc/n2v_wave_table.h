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

#ifdef N2V_TAB_STAMPS   // diagnostic build only (tools/lab/tab_stamps.sh): shader cycles per phase, summed over waves
#define N2V_STAMP(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stamps[i] += now_ - t_last_; t_last_ = now_; } while (0)
#else
#define N2V_STAMP(i) do { } while (0)
#endif

// Per-wave LDS work area.
//   feed: 64 doubles.  The left-to-right sum consumes its operands from here by broadcast reads (every lane reads the
//         same address: one LDS cycle, one instruction per operand) instead of two v_readlane per double —
//         profiles/r03: the readlane-fed sum cost 106-124 wave-cycles per slot, the LDS-fed one 41-43.  (The pairing
//         stays register-fed: an LDS read inside its dependent chain cost more latency than two v_readlane cost issue.)
//   row : the sorted row of `src` (undirected graphs), so that has_edge(nbr, src) is a binary search in LDS instead of
//         log2(deg(src)) dependent global loads per slot (297 wave-cycles per slot for tables in LDS);
//         row_n < 0: not cached (row longer than kRowCache, directed graph, first step) -> search in global memory.
constexpr int kFeed = 64;
constexpr int kRowCache = 384;     // with 8 KiB of table slots and the feed: 10 KiB per wave, 4 workgroups per CU
struct WaveScratch {
    double* feed;          // LDS [kFeed]
    const int32_t* row;    // LDS [row_n] or nullptr
    int row_n;
};

// fills ws_row[0..S) with row(src) when it fits; returns the row_n to pass on (all lanes call it)
__device__ __forceinline__ int wave_cache_row(const RowCtx& a, int32_t* ws_row, int32_t src, int lane, int cap = kRowCache) {
    if (src < 0 || !a.symmetric) return -1;
    const int64_t b = uni64(a.row_ptr[src]);
    const int S = uni((int)(a.row_ptr[src + 1] - b));
    if (S > cap) return -1;
    for (int i = lane; i < S; i += 64) ws_row[i] = a.col[b + i];
    wave_sync();
    return S;
}

__device__ __forceinline__ bool lds_row_contains(const int32_t* row, int n, int32_t v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (row[mid] < v) lo = mid + 1;
        else hi = mid;
    }
    return lo < n && row[lo] == v;
}

// unnormalised transition weight of neighbour k of the row at `base` for a step arriving from `src` (:142-148)
__device__ __forceinline__ double step_weight_of(const RowCtx& a, const WaveScratch& ws, int32_t src, int32_t nb, double wt) {
    if (src < 0) return wt;
    if (nb == src) return wt / a.p;
    const bool adj = ws.row_n >= 0 ? lds_row_contains(ws.row, ws.row_n, nb)
                     : a.symmetric ? row_contains(a.row_ptr, a.col, src, nb) : row_contains(a.row_ptr, a.col, nb, src);
    return adj ? wt : wt / a.q;
}
__device__ __forceinline__ double step_weight(const RowCtx& a, const WaveScratch& ws, int32_t src, int64_t base, int k) {
    return step_weight_of(a, ws, src, a.col[base + k], a.w ? a.w[base + k] : 1.0);
}

// norm = norm + v[0] + v[1] + ... strictly left to right (:149): the wave's 64 values are parked in LDS and added in
// order by every lane alike (broadcast reads); `cnt` of them count
__device__ __forceinline__ double wave_sum_in_order(const WaveScratch& ws, double norm, double v, int cnt, int lane) {
    ws.feed[lane] = v;
    wave_sync();
    int j = 0;
    for (; j + 8 <= cnt; j += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) norm = norm + ws.feed[j + i];
    }
    for (; j < cnt; ++j) norm = norm + ws.feed[j];
    wave_sync();
    return norm;
}

// ---- Vose's pairing (:259-268) as a two-pointer sweep ----------------------------------------------------------
// The reference pops one index from each stack, sets J[small] = large, q[large] = q[large] + q[small] - 1.0 and pushes
// `large` back on the stack its new q selects.  A pushed element is always the next one popped from its stack, so: the
// CURRENT large absorbs smalls until its q drops below 1; then it is the next small (carried in registers) and the
// next large of the stream absorbs it first.  Both memory stacks only ever hold entries of the initial classification,
// popped in a fixed order and never modified before they are popped: they are streamed 64 entries at a time, one per
// lane.  What is serial is the chain of fp64 adds of one large (read through v_readlane: same order and roundings as
// the reference); the smalls a large absorbed from one buffer are finalised by ONE call of the sink.
//   Stacks: load_small(pos, idx&, q&) / load_large(pos, idx&, q&) — entry at stack position pos; `smaller` occupies
//           positions [0, ns) (top = ns-1), `larger` positions [K-nl, K) (top = K-nl).
//   Sink:   run(in_run, idx, q, J)  per lane: slot idx is final with (q, J) for the lanes with in_run;
//           one(idx, q, J)          wave-uniform: one slot is final;
//           kRest / rest(in, idx, q) slots that were never paired are final with J = 0 and the q they have (only
//           sinks that emit results need to be told).
template <typename Stacks, typename Sink>
__device__ __forceinline__ void wave_pair(const Stacks& S, Sink& out, int K, int ns, int nl, int lane) {
    int mem_s = ns, mem_l = nl;            // stack entries not yet loaded into the lane buffers
    int s_pos = 0, s_cnt = 0, l_pos = 0, l_cnt = 0;
    int si = 0, li = 0;
    double sq = 0.0, lq = 0.0;
    bool carried = false;                  // a large whose q dropped below 1: the top of `smaller`
    int cs_i = 0;
    double cs_q = 0.0;
    for (;;) {
        if (l_pos == l_cnt) {              // next <= 64 entries of `larger` (top = position K - mem_l, then upwards)
            if (mem_l == 0) break;
            const int pos = K - mem_l + lane;
            l_cnt = min(64, mem_l);
            if (lane < l_cnt) S.load_large(pos, li, lq);
            mem_l -= l_cnt;
            l_pos = 0;
        }
        const int large = __builtin_amdgcn_readlane(li, l_pos);
        double ql = readlane_f64(lq, l_pos);       // every value of the chain is wave-uniform (v_readlane results)
        ++l_pos;
        if (carried) {                     // smaller.pop() is the element the previous large became
            out.one(cs_i, cs_q, large);    // :263
            carried = false;
            ql = ql + cs_q;                // :264, left to right
            ql = ql - 1.0;
            if (ql < 1.0) { carried = true; cs_i = large; cs_q = ql; continue; }
        }
        bool dry = false;
        for (;;) {
            if (s_pos == s_cnt) {          // next <= 64 entries of `smaller`, in pop order (top = position mem_s - 1)
                if (mem_s == 0) { dry = true; break; }
                const int pos = mem_s - 1 - lane;
                s_cnt = min(64, mem_s);
                if (lane < s_cnt) S.load_small(pos, si, sq);
                mem_s -= s_cnt;
                s_pos = 0;
            }
            const int first = s_pos;
            // The chain of one large over the buffered smalls, hand-scheduled (hipcc's version of this loop is 15
            // instructions per small, most of them re-deriving exit masks on the scalar unit; this is 10): VCC doubles
            // as the 64-bit scalar operand the two v_readlane fill.  Same operations, same order, same roundings:
            //   ql = (ql + q[s_pos]) - 1.0; ++s_pos; stop when ql < 1.0 (demoted) or the buffer is used up.
            int dem;
            {
                const int sq_lo = __double2loint(sq), sq_hi = __double2hiint(sq);
                int pos = uni(s_pos);
                const int cnt = uni(s_cnt);
                asm volatile(
                    "s_nop 3\n\t"                                  // %[pos] may come from a v_readfirstlane: lane-select hazard
                    "1:\n\t"
                    "v_readlane_b32 vcc_lo, %[qlo], %[pos]\n\t"
                    "v_readlane_b32 vcc_hi, %[qhi], %[pos]\n\t"
                    "s_add_i32 %[pos], %[pos], 1\n\t"
                    "s_nop 0\n\t"                                  // VALU-written SGPR -> VALU operand: two wait states
                    "v_add_f64 %[ql], %[ql], vcc\n\t"
                    "v_add_f64 %[ql], %[ql], -1.0\n\t"
                    "v_cmp_gt_f64 vcc, 1.0, %[ql]\n\t"
                    "s_cbranch_vccnz 2f\n\t"
                    "s_cmp_lt_i32 %[pos], %[cnt]\n\t"
                    "s_cbranch_scc1 1b\n\t"
                    "s_mov_b32 %[dem], 0\n\t"
                    "s_branch 3f\n\t"
                    "2:\n\t"
                    "s_mov_b32 %[dem], 1\n\t"
                    "3:\n\t"
                    : [ql] "+v"(ql), [pos] "+s"(pos), [dem] "=s"(dem)
                    : [qlo] "v"(sq_lo), [qhi] "v"(sq_hi), [cnt] "s"(cnt)
                    : "vcc", "scc");
                s_pos = pos;
            }
            const bool demoted = dem != 0;
            out.run(lane >= first && lane < s_pos, si, sq, large);         // :263 for every small of this run
            if (demoted) { carried = true; cs_i = large; cs_q = ql; break; }
        }
        if (dry) {                         // `smaller` is empty: the large stays on `larger` with its current q
            out.one(large, ql, 0);
            break;
        }
    }
    if (carried) out.one(cs_i, cs_q, 0);
    if (Sink::kRest) {                     // never popped: J stays 0 (:248), q as classified
        out.rest(lane >= s_pos && lane < s_cnt, si, sq);
        while (mem_s > 0) {
            const int pos = mem_s - 1 - lane;
            const int cnt = min(64, mem_s);
            if (lane < cnt) S.load_small(pos, si, sq);
            out.rest(lane < cnt, si, sq);
            mem_s -= cnt;
        }
        out.rest(lane >= l_pos && lane < l_cnt, li, lq);
        while (mem_l > 0) {
            const int pos = K - mem_l + lane;
            const int cnt = min(64, mem_l);
            if (lane < cnt) S.load_large(pos, li, lq);
            out.rest(lane < cnt, li, lq);
            mem_l -= cnt;
        }
    }
}

// ---- table in one array of {q, J, aux} slots (LDS, or global memory for the on-the-fly walk's hub rows) ---------
template <typename Slot>
struct AosTable {
    Slot* T;
    __device__ __forceinline__ void load_small(int pos, int& idx, double& q) const { idx = T[pos].aux; q = T[idx].q; }
    __device__ __forceinline__ void load_large(int pos, int& idx, double& q) const { idx = T[pos].aux; q = T[idx].q; }
};
template <typename Slot>
struct AosSink {                           // q of a stream entry is already in place: only J, and q of demoted larges
    static constexpr bool kRest = false;
    Slot* T;
    int lane;
    __device__ __forceinline__ void run(bool in_run, int idx, double, int J) const { if (in_run) T[idx].J = J; }
    __device__ __forceinline__ void one(int idx, double q, int J) const { if (lane == 0) { T[idx].q = q; T[idx].J = J; } }
    __device__ __forceinline__ void rest(bool, int, double) const {}
};

// Builds the alias table of the step that arrives at the node whose row starts at `base` (K neighbours) from `src`
// (src < 0: the first step, i.e. the node table of src/node2vec.py:184-188) in T[0..K): afterwards T[k].q / T[k].J
// are q[k] / J[k] of alias_setup.  All 64 lanes of the wave call it together.  Returns false when the weights sum
// to 0 (the reference raises ZeroDivisionError, :150 / :187).
template <typename Slot>
__device__ __forceinline__ bool wave_build_table(const RowCtx& a, Slot* T, const WaveScratch ws, int32_t src, int64_t base,
                                                 int K, int lane, unsigned long long* stamps = nullptr) {
#ifdef N2V_TAB_STAMPS
    unsigned long long t_last_ = __builtin_amdgcn_s_memtime();
#endif
    // ---- 1. unnormalised weights in parallel (:142-148)
    for (int k = lane; k < K; k += 64) T[k].q = step_weight(a, ws, src, base, k);
    wave_sync();
    N2V_STAMP(0);
    // ---- 2. norm = sum(unnormalized_probs), strictly left to right (:149)
    double norm = 0.0;
    for (int c = 0; c < K; c += 64)
        norm = wave_sum_in_order(ws, norm, (c + lane < K) ? T[c + lane].q : 0.0, min(64, K - c), lane);
    norm = unid(norm);
    N2V_STAMP(1);
    if (norm == 0.0) return false;
    // ---- 3. q = K * (u / norm) (:150 then :253, two roundings) and the two index stacks in index order
    //         (:252-257): `smaller` grows up from position 0, `larger` down from position K-1
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
    N2V_STAMP(2);
    // ---- 4. pairing (:259-268)
    AosSink<Slot> sink{T, lane};
    wave_pair(AosTable<Slot>{T}, sink, K, ns, nl, lane);
    wave_sync();
    N2V_STAMP(3);
    return true;
}

// ---- draw before building (on-the-fly walk) ------------------------------------------------------------------------
// alias_setup only ever rewrites q[large] (:264): a slot it classifies as `smaller` (q = K * prob < 1, :253-255) keeps
// that q for good.  When the step's first uniform lands on such a slot kk and the second accepts it (u2 < q[kk], :278),
// alias_draw returns kk without looking at J or at any other slot: the step needs the K weights, their left-to-right
// sum and ONE division — not the classification, the stacks or the pairing (54-60 % of a table's cycles, tab_stamps).
// wave_weights_and_norm = phases 1-2 of wave_build_table; wave_finish_table = phases 3-4.
template <typename Slot>
__device__ __forceinline__ bool wave_weights_and_norm(const RowCtx& a, Slot* T, const WaveScratch ws, int32_t src, int64_t base,
                                                      int K, int lane, double& norm_out) {
    for (int k = lane; k < K; k += 64) T[k].q = step_weight(a, ws, src, base, k);
    wave_sync();
    double norm = 0.0;
    for (int c = 0; c < K; c += 64)
        norm = wave_sum_in_order(ws, norm, (c + lane < K) ? T[c + lane].q : 0.0, min(64, K - c), lane);
    norm_out = unid(norm);
    return norm_out != 0.0;
}
template <typename Slot>
__device__ __forceinline__ void wave_finish_table(Slot* T, int K, double norm, int lane) {
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
    AosSink<Slot> sink{T, lane};
    wave_pair(AosTable<Slot>{T}, sink, K, ns, nl, lane);
    wave_sync();
}

// Unweighted undirected graphs whose 1/p and 1/q are dyadic (multiples of 2^-20 up to 2^10; degrees below 2^21): every
// weight is one of {1/p, 1, 1/q}, every partial sum of the reference's left-to-right sum (:149) is a multiple of 2^-20
// below 2^32 — exact in fp64 — so the sum does not depend on the order and equals
//     [prev in row(cur)] * 1/p  +  n_adj * 1  +  (K - [prev in row(cur)] - n_adj) * 1/q,     each product and sum exact,
// with n_adj = |{x in row(cur): x != prev, x in row(prev)}| (has_edge(x, prev) on an undirected graph, :145).  The count
// walks the SHORTER of the two sorted rows and searches the longer one (dyadic_draw below); no serial chain.
// A long sorted row searched by the lanes of one wave.  Binary searches in global memory are log2(n) DEPENDENT loads each
// (a hub row of thousands: ~8 us); here the row is first copied into `big` (LDS, `cap` ints) with pipelined coalesced
// loads when it fits, else every stride-th element is (pivots): a search is LDS probes plus at most log2(stride)
// dependent loads inside one segment.
struct StagedRow {
    const int32_t* g;      // the row in global memory
    const int32_t* big;    // LDS: the row (stride 1) or its pivots g[0], g[stride], g[2 stride], ...
    int n, stride, n_big;
    __device__ __forceinline__ int find(int32_t v) const {   // position of v in the row, -1 if absent
        int lo = 0, hi = n_big;                       // first staged entry >= v
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (big[mid] < v) lo = mid + 1;
            else hi = mid;
        }
        if (lo < n_big && big[lo] == v) return lo * stride;
        if (stride == 1 || lo == 0) return -1;
        const int top = min(lo * stride, n);
        int glo = (lo - 1) * stride + 1, ghi = top;   // strictly between pivots lo-1 and lo
        while (glo < ghi) {
            const int mid = (glo + ghi) >> 1;
            if (g[mid] < v) glo = mid + 1;
            else ghi = mid;
        }
        return (glo < top && g[glo] == v) ? glo : -1;
    }
    __device__ __forceinline__ bool contains(int32_t v) const { return find(v) >= 0; }
};
__device__ __forceinline__ StagedRow stage_row(const int32_t* __restrict__ g, int n, int32_t* big, int cap, int lane) {
    const int stride = (n + cap - 1) / cap;           // 1 when the row fits
    const int n_big = (n + stride - 1) / stride;
    for (int i = lane; i < n_big; i += 64) big[i] = g[(int64_t)i * stride];
    wave_sync();
    return StagedRow{g, big, n, stride, n_big};
}

// The whole draw of a dyadic step, any K: alias_draw's pick (:277-281) without a table.
//  1. count (above) — and note WHERE in cur's row the special slots sit: prev's (weight 1/p) and the common neighbours'
//     (weight 1), ascending, in `spec` (LDS, kSpecCap ints; more common neighbours than that: returns -2 and the caller
//     builds the table).  Every other slot weighs 1/q.
//  2. q takes three values, one per class — K * (w_class / norm), the builder's two roundings — so a class is `smaller`
//     or `larger` as a whole; if slot kk is smaller and u2 < q it is the pick (98 % of the draws on a hub row).
//  3. otherwise Vose's sweep (:259-268) is replayed on the CLASSES: both stacks were pushed in index order, so each is a
//     descending walk over the indices of its classes (jumping from special to special when the 1/q class is on the other
//     stack); the chain of fp64 adds is the reference's, operand by operand; it stops when slot kk is final, as in
//     wave_draw_le64.  No table, no stacks, no memory traffic beyond `spec`: a rejected draw on a row of thousands costs
//     ~20 us instead of the ~240 us of building its table.
// ws.row / ws.row_n: prev's row staged in LDS (wave_cache_row) or row_n < 0.  big/cap: LDS scratch for the longer row
// (the table window, unused until a build).
constexpr int kSpecCap = 128;
struct ClassSweep {
    const int32_t* M;      // the special slots, ascending: position * 4 + class (1 = weight 1, 2 = 1/p); every other slot: class 0 (1/q)
    int m;
    int smask;             // bit cl: class cl is `smaller`
    double qv;             // lane cl holds the class's q (read by v_readlane: no indexed memory)
    __device__ __forceinline__ bool sm(int cl) const { return (smask >> cl) & 1; }
    __device__ __forceinline__ double qc(int cl) const { return readlane_f64(qv, uni(cl)); }
};
// One of Vose's two stacks as a descending sequence of RUNS: a special slot (one index) or the gap of 1/q slots between
// two specials (all of one class, so all on the same stack) — a hub row of thousands is a handful of runs.
struct RunIter {
    int j, top;            // next special (index into M, descending), highest index not yet covered
    int want;              // 1: the `smaller` stack, 0: the `larger` one
    int r_top, r_cnt, r_cls;   // current run: indices r_top, r_top-1, ... (r_cnt of them; 0 = the stack is empty)
    __device__ __forceinline__ void next(const ClassSweep& c) {
        for (;;) {
            if (top < 0) { r_cnt = 0; return; }
            const int e = j >= 0 ? uni(c.M[j]) : -4;
            const int spos = e >> 2;
            if (spos == top) {
                --j;
                r_top = top; r_cnt = 1; r_cls = e & 3;
                --top;
            } else {
                r_top = top; r_cnt = top - spos; r_cls = 0;
                top = spos;
            }
            if ((int)c.sm(r_cls) == want) return;
        }
    }
    __device__ __forceinline__ void take(const ClassSweep& c, int n) {   // n <= r_cnt entries popped
        r_top -= n;
        r_cnt -= n;
        if (r_cnt == 0) next(c);
    }
};
// exact_sum false (1/p or 1/q not dyadic; still an unweighted undirected graph, still three weight classes): the sum is
// the reference's left-to-right chain (:149), run by run — a gap of n far slots is n times `norm = norm + 1/q` in
// registers — instead of the count formula; everything else is the same.
__device__ __forceinline__ int dyadic_draw(const RowCtx& a, const WaveScratch& ws, int32_t prev, int64_t base, int K, int kk,
                                           double u2, double wp, double wq, bool exact_sum, int32_t* big, int cap, int32_t* spec,
                                           int lane) {
    if (prev < 0) {                          // first step: the node table, all weights 1 (:184-188): one class, no pairing
        const double q = (double)K * (1.0 / (double)K);
        return (u2 < q) ? kk : 0;
    }
    const int64_t pb = uni64(a.row_ptr[prev]);
    const int S = uni((int)(a.row_ptr[prev + 1] - pb));
    const int32_t* rc = a.col + base;
    const int32_t* rp = a.col + pb;
    const bool cached = ws.row_n >= 0;       // == S
    int n_adj = 0, p_idx = -1;
    const unsigned long long below = (1ULL << lane) - 1ULL;
    auto note = [&](bool adj, int pos) {     // append the positions of this chunk's common neighbours (ascending already)
        const unsigned long long m = __ballot(adj);
        const int at = n_adj + __popcll(m & below);
        if (adj && at < kSpecCap) spec[at] = pos;     // (beyond the list: counted, and the caller builds the table)
        n_adj += __popcll(m);
    };
    if ((cached && K <= 8 * S) || K <= S) {  // walk cur's row, search prev's (its LDS copy, or staged now)
        StagedRow R{rp, ws.row, S, 1, S};
        if (!cached) R = stage_row(rp, S, big, cap, lane);
        for (int c = 0; c < K; c += 64) {
            const int k = c + lane;
            bool adj = false, isp = false;
            if (k < K) {
                const int32_t nb = rc[k];
                isp = nb == prev;
                adj = !isp && R.contains(nb);
            }
            note(adj, k);
            const unsigned long long mp = __ballot(isp);
            if (mp != 0ULL) p_idx = c + __builtin_ctzll(mp);
        }
    } else {                                 // walk prev's row, find its entries in cur's long row
        const StagedRow R = stage_row(rc, K, big, cap, lane);
        for (int c = 0; c < S; c += 64) {
            const int i = c + lane;
            int pos = -1;
            if (i < S) {
                const int32_t x = cached ? ws.row[i] : rp[i];
                if (x != prev) pos = R.find(x);
            }
            note(pos >= 0, pos);
        }
        p_idx = R.find(prev);
    }
    n_adj = uni(n_adj);
    p_idx = uni(p_idx);
    wave_sync();                             // spec is read below; `big` is the table window: done with it
    const int n_prev = p_idx >= 0 ? 1 : 0;
    const int m = n_adj + n_prev;
    if (m > kSpecCap) return -2;
    // spec -> M: tag the common neighbours with class 1 and insert prev's slot (class 2) at its place
    {
        const int v0 = lane < n_adj ? spec[lane] : 0x3FFFFFFF, v1 = 64 + lane < n_adj ? spec[64 + lane] : 0x3FFFFFFF;
        const int before = __popcll(__ballot(v0 < p_idx)) + __popcll(__ballot(v1 < p_idx));   // p_idx = -1: 0
        wave_sync();
        if (lane < n_adj) spec[lane + ((n_prev && v0 > p_idx) ? 1 : 0)] = v0 * 4 + 1;
        if (64 + lane < n_adj) spec[64 + lane + ((n_prev && v1 > p_idx) ? 1 : 0)] = v1 * 4 + 1;
        if (n_prev && lane == 0) spec[before] = p_idx * 4 + 2;
        wave_sync();
    }
    double norm;
    if (exact_sum) {
        norm = ((double)n_prev * wp + (double)n_adj) + (double)(K - n_prev - n_adj) * wq;   // exact (see above)
    } else {                                 // :149, left to right over the runs
        norm = 0.0;
        int pos = 0;
        for (int j = 0; j <= m; ++j) {
            const int e = j < m ? uni(spec[j]) : K * 4;
            for (int g = (e >> 2) - pos; g > 0; --g) norm = norm + wq;
            if (j < m) norm = norm + ((e & 3) == 2 ? wp : 1.0);
            pos = (e >> 2) + 1;
        }
        norm = unid(norm);
    }
    if (norm == 0.0) return -2;              // (negative p or q) the table builder reports it (:150)
    const double Kd = (double)K;
    const double q0 = Kd * (wq / norm), q1 = Kd * (1.0 / norm), q2 = Kd * (wp / norm);
    const ClassSweep c{spec, m, (q0 < 1.0 ? 1 : 0) | (q1 < 1.0 ? 2 : 0) | (q2 < 1.0 ? 4 : 0),
                       lane == 0 ? q0 : (lane == 1 ? q1 : q2)};
    int cls_kk = 0;
    {
        const int e0 = lane < m ? spec[lane] : -4, e1 = 64 + lane < m ? spec[64 + lane] : -4;
        const unsigned long long h0 = __ballot((e0 >> 2) == kk), h1 = __ballot((e1 >> 2) == kk);
        if (h0) cls_kk = __builtin_amdgcn_readlane(e0, __builtin_ctzll(h0)) & 3;
        else if (h1) cls_kk = __builtin_amdgcn_readlane(e1, __builtin_ctzll(h1)) & 3;
    }
    double q_kk = c.qc(cls_kk);
    if (q_kk < 1.0 && u2 < q_kk) return kk;  // `smaller` and accepted (:278)
    RunIter sm_{m - 1, K - 1, 1, 0, 0, 0}, lg_{m - 1, K - 1, 0, 0, 0, 0};
    sm_.next(c);
    lg_.next(c);
    int J_kk = 0, c_i = 0;
    bool carried = false, done = false;
    double c_q = 0.0;
    while (!done && lg_.r_cnt > 0 && (carried || sm_.r_cnt > 0)) {        // :259
        const int large = lg_.r_top;                                    // larger.pop()
        double ql = c.qc(lg_.r_cls);
        lg_.take(c, 1);
        for (;;) {
            if (carried) {                                            // smaller.pop() is the large that dropped below 1
                carried = false;
                if (c_i == kk) { J_kk = large; q_kk = c_q; done = true; break; }
                ql = ql + c_q;                                        // :264, left to right
                ql = ql - 1.0;
            } else {
                if (sm_.r_cnt == 0) { if (large == kk) q_kk = ql; done = true; break; }   // `smaller` is dry
                const double qs = c.qc(sm_.r_cls);
                int n = sm_.r_cnt;                                      // smalls of this run popped before slot kk is
                const bool hit = kk <= sm_.r_top && kk > sm_.r_top - n;
                if (hit) n = sm_.r_top - kk;
                int used = 0;
                bool dem = false;
                while (used < n) {                                    // the chain of one large over a run of equal smalls
                    ql = ql + qs;
                    ql = ql - 1.0;
                    ++used;
                    if (uni((int)(ql < 1.0))) { dem = true; break; }
                }
                sm_.take(c, used);
                if (!dem) {
                    if (hit) { J_kk = large; q_kk = qs; done = true; break; }   // :263 — the next small is slot kk
                    continue;
                }
            }
            if (uni((int)(ql < 1.0))) {                               // :265-266 — the large is the next small
                if (lg_.r_cnt == 0) { if (large == kk) q_kk = ql; done = true; }  // nothing left to absorb it: J stays 0
                else { carried = true; c_i = large; c_q = ql; }
                break;
            }
        }
    }
    return (u2 < q_kk) ? kk : J_kk;                                   // :278-281
}

// Rows of at most 64 neighbours (84 % of the steps of a walk on a power-law graph, and nearly all whose slot kk is NOT
// accepted at once): the whole table lives in registers, one slot per lane — no LDS table, no index stacks.  The two
// stacks are two lane masks (both were pushed in index order, so the top of either is its highest set bit, :252-257,
// 260-261), a slot's q is a v_readlane away, and the sweep stops as soon as slot kk is final: a small is final when it
// is popped (:263 gives its J; its q never changed), a large when `smaller` runs dry or — after it dropped below 1 —
// when the next large absorbs it; slots the sweep never reaches keep q and J = 0 (:248).  Returns the slot alias_draw
// picks (:277-281) or -1 when the weights sum to 0 (:150, ZeroDivisionError).
//   exact_sum: every partial sum is exact (dyadic weights, see above) — the butterfly sum equals the left-to-right one.
//   nb: the lane's neighbour col[base + lane] (lanes < K), already loaded by the caller.
//   wp, wq: 1/p, 1/q (used when exact_sum: the weight classes are lane masks and the sum is a count, as in dyadic_draw).
// Slot kk is tested BEFORE the other lanes normalise (one fp64 division instead of 64 lanes' worth of them).
__device__ __forceinline__ int wave_draw_le64(const RowCtx& a, const WaveScratch& ws, int32_t src, int64_t base, int32_t nb, int K,
                                              int kk, double u2, bool exact_sum, double wp, double wq, int lane) {
    const bool valid = lane < K;
    double w, norm, w_kk;
    if (exact_sum) {
        const bool isp = valid && src >= 0 && nb == src;
        const bool adj = valid && src >= 0 && !isp &&
                         (ws.row_n >= 0 ? lds_row_contains(ws.row, ws.row_n, nb) : row_contains(a.row_ptr, a.col, src, nb));
        const unsigned long long mp = __ballot(isp), ma = __ballot(adj);
        if (src < 0) { norm = (double)K; w_kk = 1.0; w = 1.0; }      // the node table: all weights 1 (:184-188)
        else {
            const int n_prev = __popcll(mp), n_adj = __popcll(ma);
            norm = ((double)n_prev * wp + (double)n_adj) + (double)(K - n_prev - n_adj) * wq;   // exact, see dyadic_draw
            w_kk = ((mp >> kk) & 1ULL) ? wp : (((ma >> kk) & 1ULL) ? 1.0 : wq);
            w = isp ? wp : (adj ? 1.0 : wq);
        }
    } else {
        w = valid ? step_weight_of(a, ws, src, nb, a.w ? a.w[base + lane] : 1.0) : 0.0;
        norm = unid(wave_sum_in_order(ws, 0.0, w, K, lane));
        if (norm == 0.0) return -1;
        w_kk = readlane_f64(w, kk);
    }
    double q_kk = (double)K * (w_kk / norm);                     // :150 then :253
    if (q_kk < 1.0 && u2 < q_kk) return kk;                      // `smaller` and accepted: nothing else matters
    const double q = valid ? (double)K * (w / norm) : 0.0;
    unsigned long long ms = __ballot(valid && q < 1.0), ml = __ballot(valid && !(q < 1.0));
    int J_kk = 0;
    bool carried = false, done = false;
    int c_i = 0;
    double c_q = 0.0;
    while (!done && ml != 0ULL && (carried || ms != 0ULL)) {     // :259
        const int large = 63 - __builtin_clzll(ml);               // larger.pop()
        ml &= ~(1ULL << large);
        double ql = readlane_f64(q, large);
        for (;;) {
            int small;
            double qs;
            if (carried) { small = c_i; qs = c_q; carried = false; }
            else if (ms != 0ULL) { small = 63 - __builtin_clzll(ms); ms &= ~(1ULL << small); qs = readlane_f64(q, small); }
            else {                                                // `smaller` is dry: the large stays with its current q
                if (large == kk) q_kk = ql;
                done = true;
                break;
            }
            if (small == kk) { J_kk = large; q_kk = qs; done = true; break; }   // :263 — slot kk is final
            ql = ql + qs;                                         // :264, left to right
            ql = ql - 1.0;
            if (uni((int)(ql < 1.0))) {                           // :265-266 — the large is the next small
                if (ml == 0ULL) { if (large == kk) q_kk = ql; done = true; }    // nothing left to absorb it: J stays 0
                else { carried = true; c_i = large; c_q = ql; }
                break;
            }
            if (large == kk && ms == 0ULL) { q_kk = ql; done = true; break; }   // back on `larger`, `smaller` dry
        }
    }
    return (u2 < q_kk) ? kk : J_kk;                               // :278-281
}

// ---- large tables of the table BUILDER: nothing but the two stacks is stored ---------------------------------------
// A table too large for the wave's LDS slots used to be built in place in its output (q, J and the stack words landed
// as partial-line writes in the 32-B fat slots before the final slot was written: 2.65x the output in write traffic on
// C3, profiles/r02).  Here the unnormalised weights are computed twice (once for the sum, once to classify — a column
// load and a search in the LDS-cached source row) instead of being stored; the classification appends {index, q} to
// the two stacks, which live in a per-wave scratch and are written and read once, sequentially; the pairing hands every
// finished slot (index, q, J) to the emitter, which queues 64 of them in LDS and then writes 64 complete output slots.
struct StreamStacks {                      // per-wave global scratch: idx[K], q[K]
    int32_t* idx;
    double* q;
    __device__ __forceinline__ void load_small(int pos, int& i, double& v) const { i = idx[pos]; v = q[pos]; }
    __device__ __forceinline__ void load_large(int pos, int& i, double& v) const { i = idx[pos]; v = q[pos]; }
};

// Emit: void operator()(bool active, int idx, double q, int J) — called by all 64 lanes, writes the output slot of idx.
template <typename Emit>
struct QueueSink {
    static constexpr bool kRest = true;
    int32_t* qi;                           // LDS [128] slot index
    int32_t* qJ;                           // LDS [128]
    double* qq;                            // LDS [128]
    Emit emit;
    int lane;
    int n = 0;                             // queued entries, < 64 between calls
    __device__ __forceinline__ void flush64() {
        wave_sync();
        emit(true, qi[lane], qq[lane], qJ[lane]);
        wave_sync();
        const int rem = n - 64;            // < 64: move the tail to the front
        if (lane < rem) { const int i = qi[64 + lane], J = qJ[64 + lane]; const double q = qq[64 + lane];
                          qi[lane] = i; qJ[lane] = J; qq[lane] = q; }
        n = rem;
    }
    __device__ __forceinline__ void run(bool in_run, int idx, double q, int J) {
        const unsigned long long m = __ballot(in_run);
        if (in_run) { const int p = n + __popcll(m & ((1ULL << lane) - 1ULL)); qi[p] = idx; qJ[p] = J; qq[p] = q; }
        n += __popcll(m);
        if (n >= 64) flush64();
    }
    __device__ __forceinline__ void one(int idx, double q, int J) {
        if (lane == 0) { qi[n] = idx; qJ[n] = J; qq[n] = q; }
        if (++n >= 64) flush64();
    }
    __device__ __forceinline__ void rest(bool in, int idx, double q) { run(in, idx, q, 0); }
    __device__ __forceinline__ void finish() {
        wave_sync();
        emit(lane < n, qi[lane], qq[lane], qJ[lane]);
        n = 0;
        wave_sync();
    }
};

template <typename Emit>
__device__ __forceinline__ bool wave_build_stream(const RowCtx& a, const StreamStacks S, const WaveScratch ws,
                                                  QueueSink<Emit>& sink, int32_t src, int64_t base, int K, int lane,
                                                  unsigned long long* stamps = nullptr) {
#ifdef N2V_TAB_STAMPS
    unsigned long long t_last_ = __builtin_amdgcn_s_memtime();
#endif
    // ---- 1 + 2. weights (:142-148) straight into the left-to-right sum (:149); nothing is stored
    double norm = 0.0;
    for (int c = 0; c < K; c += 64)
        norm = wave_sum_in_order(ws, norm, (c + lane < K) ? step_weight(a, ws, src, base, c + lane) : 0.0, min(64, K - c), lane);
    norm = unid(norm);
    N2V_STAMP(1);
    if (norm == 0.0) return false;
    // ---- 3. the same weights again, q = K * (u / norm) (:150, :253), {k, q} onto the stack q selects (:252-257)
    const double Kd = (double)K;
    int ns = 0, nl = 0;
    for (int c = 0; c < K; c += 64) {
        const int k = c + lane;
        const bool valid = k < K;
        double qk = 0.0;
        if (valid) qk = Kd * (step_weight(a, ws, src, base, k) / norm);
        const bool is_small = valid && (qk < 1.0);
        const unsigned long long ms = __ballot(is_small), ml = __ballot(valid && !is_small);
        const unsigned long long below = (1ULL << lane) - 1ULL;
        if (valid) {
            const int pos = is_small ? ns + __popcll(ms & below) : K - (nl + __popcll(ml & below) + 1);
            S.idx[pos] = k;
            S.q[pos] = qk;
        }
        ns += __popcll(ms);
        nl += __popcll(ml);
    }
    ns = uni(ns);
    nl = uni(nl);
    wave_sync();
    N2V_STAMP(2);
    // ---- 4. pairing (:259-268); finished slots leave through the queue
    wave_pair(S, sink, K, ns, nl, lane);
    sink.finish();
    N2V_STAMP(3);
    return true;
}

}  // namespace n2v
