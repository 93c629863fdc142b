"""Replica merges of the multi-GPU skip-gram trainer: how G replicas of syn0 / syn1neg — one per GPU,
each trained on its shard of the walks by n2v_hip.sgns — are combined over RCCL.  Two schemes:

  * merge="tsum" (default), SumTierPlan / TieredSumMerger: every row's changes are SUMMED over the ranks at a cadence
    that depends on the row — all rows a few hundred times per pass, hub rows 4 / 16 / 64 times as often;
  * merge="hot", MergePlan / ReplicaMerger: per-row weights between sum and mean at auto_syncs merges per pass.

The reference has no counterpart (gensim's worker threads share ONE table, src/main.py:87), so only the
link-prediction AUC band judges a scheme; the measurements behind the constants are below and in DESIGN.md 6.
The arithmetic runs in csrc/n2v_merge.hip (HipMergeOps); the communicator is whatever offers all_reduce_async
(n2v_hip.dist._Comm over RCCL, gloo in the CPU tests)."""
import ctypes

import numpy as np
import torch

from . import _lib


def shard_bounds(n_items, world, rank):
    """Contiguous shard [begin, end) of n_items for `rank` (as src/main_link.py:261-264 splits
    start nodes among its pool workers); also the row ranges of a pipelined merge."""
    per = -(-n_items // world)
    b = min(rank * per, n_items)
    return b, min(b + per, n_items)

# ------------------------------------------------------------------------------------------- replica merges
# How G replicas (one per GPU, each trained on its shard of the walks) are combined.  Measured on one MI355X by
# training G replicas interval by interval (tests/probes/merge_lab.py; sequential single-thread CPU comparator on the
# same walks; 3 000-node uniform graph, comparator AUC 0.89607, and 20 000-node hub graph, max degree 597, comparator
# 0.86678; logs under profiles/r02/logs/merge_lab*.log; a configuration reproduces to 1e-4 from run to run):
#   * summing the replicas' changes ("delta") is what one shared Hogwild table would have received, but only for
#     rows that get a handful of updates per interval: hub rows and frequent negatives run into saturation inside
#     every replica and the sum then overshoots by the factor G (hub graph, 8 replicas: +0.03 at 234 merges per
#     pass, divergence at 59);
#   * the mean ("avg", local SGD) under-trains every cold row by 1/G (-0.05 at 8 replicas);
#   * 'hot': per-row weight on the sum, w = lam + (1 - lam)/G, lam = min(1, HOT_BUDGET / u), u = expected updates
#     of the row by the OTHER replicas per interval.  AUC minus comparator at STALENESS_BUDGET 48 for HOT_BUDGET
#     160 / 192 / 224 / 256 / 320: hub graph, 8 replicas -0.0011 / -0.0011 / -0.0017 / -0.0027 / -0.0068, uniform
#     graph +0.0021 / +0.0018 / +0.0014 / +0.0010 / +0.0002 (two replicas: -0.0015 and +0.0018 whatever the budget);
#     budgets of 512 and more pass through an unstable band (-0.014 ... +0.02) before they reach the pure sum.
#     208 keeps all four cases inside +-0.002, with margins of 0.0003-0.0006 — the scheme has no slack left.  The
#     best budget depends on the replica count: two replicas 512 / 768 / 1024 -> hub -0.0010 / -0.0014 / -0.0013,
#     uniform +0.0010 / +0.0005 / -0.0001; FOUR replicas are the hardest case — the hub graph wants a small budget
#     and a coarse cadence, the uniform graph the opposite: (HOT_BUDGET, STALENESS_BUDGET) = (160, 24) is the only
#     point found inside the band on both (-0.0014 / +0.0019), (208, 48) gives -0.0019 / +0.0025 (lab6-8 logs).
#     MERGE_CONSTANTS holds the per-count choices; counts in between take the nearest measured one;
#   * the cadence: STALENESS_BUDGET tokens per vocabulary row and interval from the other replicas.  24 / 32 / 48 /
#     64 / 96 at 8 replicas (HOT_BUDGET 256): hub -0.0055 / -0.0047 / -0.0027 / -0.0023 / -0.0035, uniform -0.0003 /
#     -0.0001 / +0.0010 / +0.0020 / -0.0004;
#   * smooth "contraction model" weights w = (1 - exp(-G h)) / (G (1 - exp(-h))) with h ~ updates / n0 (optionally
#     scaled by the decaying learning rate) were no better and less stable (merge_lab3.log);
#   * merging LATE — a replica keeps training while the sum of the last interval is still travelling, which is what
#     hiding the all-reduce behind the next interval's training amounts to — breaks the band at 8 replicas however
#     the late sum is applied (all rows: -0.017 hub, -0.085 uniform; only rows with u <= 256: -0.010 / -0.045;
#     own change pre-weighted: -0.008 / -0.12); with u <= 64 it is neutral but then delays almost nothing (at this
#     cadence an average row already has u ~ 500).  The delayed tier therefore exists (cold_delay=True, HOT_THETA)
#     and is tested, but is OFF by default; what is hidden instead is the merge ARITHMETIC behind the wire time:
#     the synchronous merge is pipelined over row ranges (ReplicaMerger.end_interval).
#   * LIMIT OF THE SCHEME (lab9 / lab10 logs): the constants below were fitted on the two SMALL probe graphs.  On a
#     131 072-node hub graph (10 walks of 80, 12-thread CPU comparator 0.88175, one GPU 0.88019) the same constants
#     give -0.0019 (2 replicas), -0.0047 (4) and -0.0064 (8): at fixed tokens per row and interval the bias grows with
#     the graph, every row being damped to about half of the summed change (w ~ 0.5 at u ~ 500) for the whole pass
#     while the learning rate — and with it the saturation the damping is there for — decays to zero.  Making the
#     budget grow with 1 / learning rate helps the large graph (-0.0034 / -0.0031 at 8 / 4 replicas) and hurts the
#     small hub graph (-0.0061): no member of this family is inside the band at every size.  What is enforced by the
#     tests is therefore the band on the two small graphs; C4-sized multi-GPU parity is an open item.
#   * What does reach the comparator at every size in simulation: PURE SUMS at per-row cadences (merge="tsum" below;
#     lab13-15) — every row merged 234 times per pass, rows with more than 125 expected updates by the others per base
#     interval 4 / 16 / 64 times as often: product path, 8 replicas: hub graph -0.0001, uniform -0.0000; lab, 131k-node
#     hub graph -0.0002.  An option, not the default: with one wavefront per walk the launches between two hub-tier
#     merges cover ~100 walks at C4's size (2.4e7 pairs/s per GPU, tools/sgns_launch_probe.py).
# The arithmetic around the collectives is three fused kernels (csrc/n2v_merge.hip).
HOT_BUDGET = 208.0
HOT_THETA = 64.0
STALENESS_BUDGET = 48.0
# replicas -> (HOT_BUDGET, STALENESS_BUDGET); measured at 2, 4 and 8 replicas
MERGE_CONSTANTS = {2: (512.0, 48.0), 4: (160.0, 24.0), 8: (HOT_BUDGET, STALENESS_BUDGET)}


def merge_constants(world):
    """(hot budget, staleness budget) for `world` replicas: the measured count nearest to it (ties: the larger)."""
    if world <= 1:
        return HOT_BUDGET, STALENESS_BUDGET
    key = min(MERGE_CONSTANTS, key=lambda g: (abs(np.log2(g) - np.log2(world)), -g))
    return MERGE_CONSTANTS[key]
MIN_WALKS_PER_LAUNCH = 8192  # informational: one wavefront trains one walk at a time; 5 356-walk launches still ran
                             # at the full-pass rate (tools/sgns_grid_probe.py)


def expected_updates(counts, interval_tokens_global, window, negative, device):
    """Expected updates per interval over ALL replicas of (syn0 rows, syn1neg rows): a syn0 row is the input of
    ~(window + 0.5) pairs per occurrence of its word, a syn1neg row the positive target of as many plus
    `negative` draws per pair from the unigram^0.75 distribution."""
    c = torch.as_tensor(counts, dtype=torch.float64, device=device)
    pv = c / c.sum().clamp_min(1)
    pn = c ** 0.75
    pn = pn / pn.sum().clamp_min(1e-300)
    ppt = window + 0.5
    return ppt * interval_tokens_global * pv, ppt * interval_tokens_global * (pv + negative * pn)


def merge_weights(counts, interval_tokens_global, world, window, negative, device, budget=HOT_BUDGET, with_u=False):
    """Per-row weights (w_syn0, w_syn1neg) on the SUM of the replicas' changes: with u = (world-1)/world * expected
    updates per interval, lam = min(1, budget / u) and w = lam + (1 - lam) / world.  budget = inf gives the pure
    sum, budget = 0 the mean."""
    out, us = [], []
    for upd in expected_updates(counts, interval_tokens_global, window, negative, device):
        u = (world - 1) / world * upd
        lam = torch.clamp(budget / u.clamp_min(1e-30), max=1.0) if budget > 0 else torch.zeros_like(u)
        us.append(u)
        out.append((lam + (1 - lam) / world).to(torch.float32))
    return (out, us) if with_u else out


class MergePlan:
    """Weights and tiers of one run: identical on every rank (derived from the global word counts)."""

    def __init__(self, counts, interval_tokens_global, world, window, negative, device, mode="hot",
                 budget=None, theta=HOT_THETA, cold_delay=False):
        if mode not in ("hot", "delta", "avg"):
            raise ValueError("merge mode %r" % (mode,))
        if budget is None:
            budget = merge_constants(world)[0]
        b = {"hot": budget, "delta": float("inf"), "avg": 0.0}[mode]
        self.w, us = merge_weights(counts, interval_tokens_global, world, window, negative, device, b, with_u=True)
        self.world, self.cold_delay = world, bool(cold_delay)
        self.hot_rows, self.hot_pos = [], []
        for u in us:
            hot = (u > theta) if cold_delay else torch.ones_like(u, dtype=torch.bool)
            rows = torch.nonzero(hot).flatten()
            pos = torch.full((u.numel(),), -1, dtype=torch.int32, device=device)
            pos[rows] = torch.arange(rows.numel(), dtype=torch.int32, device=device)
            self.hot_rows.append(rows.contiguous())
            self.hot_pos.append(pos)
        self.n_hot = [int(r.numel()) for r in self.hot_rows]
        self.n_cold = [int(u.numel()) - h for u, h in zip(us, self.n_hot)]


class HipMergeOps:
    """The merge arithmetic on device tensors: csrc/n2v_merge.hip.  No CPU path."""

    def __init__(self):
        self.lib = _lib.load()

    @staticmethod
    def _bf16(t):
        if t.dtype == torch.bfloat16:
            return 1
        if t.dtype == torch.float32:
            return 0
        raise TypeError("wire buffers are bfloat16 or float32, not %s" % (t.dtype,))

    def _check(self, x):
        if not x.is_cuda:
            raise RuntimeError("n2v_hip: replica merges run on the GPU; there is no CPU fallback")

    def snapshot(self, x, xs, base, w, hot_pos, sum_prev, cold_wire, hot_wire):
        self._check(x)
        ref = cold_wire if cold_wire is not None else hot_wire
        with torch.cuda.device(x.device):
            _lib.check(self.lib.n2v_merge_snapshot(
                _lib.ptr(x), _lib.ptr(xs), _lib.ptr(base), int(x.shape[0]), int(x.shape[1]), _lib.ptr(w),
                _lib.ptr(hot_pos), _lib.ptr(sum_prev), _lib.ptr(cold_wire), _lib.ptr(hot_wire), self._bf16(ref),
                _lib.stream_ptr(x.device)))

    def hot_apply(self, x, xs, base, w, hot_rows, hot_sum):
        self._check(x)
        with torch.cuda.device(x.device):
            _lib.check(self.lib.n2v_merge_hot_apply(
                _lib.ptr(x), _lib.ptr(xs), _lib.ptr(base), int(x.shape[1]), _lib.ptr(w), _lib.ptr(hot_rows),
                int(hot_rows.numel()), _lib.ptr(hot_sum), self._bf16(hot_sum), _lib.stream_ptr(x.device)))

    def pack_rows(self, x, base, rows, wire):
        self._check(x)
        with torch.cuda.device(x.device):
            _lib.check(self.lib.n2v_merge_pack_rows(
                _lib.ptr(x), _lib.ptr(base), int(x.shape[1]), _lib.ptr(rows), int(rows.numel()), _lib.ptr(wire),
                self._bf16(wire), _lib.stream_ptr(x.device)))

    class _TsumTable(ctypes.Structure):       # n2v_tsum_table of include/n2v_hip.h
        _fields_ = [("table", ctypes.c_void_p), ("base", ctypes.c_void_p), ("rows", ctypes.c_void_p),
                    ("n_rows", ctypes.c_int64)]

    def tsum_tables(self, tables, bases, row_lists):
        """The n2v_tsum_table array of one merge level (row_lists[i] None: every row of table i).  The caller keeps
        the tensors alive."""
        arr = (self._TsumTable * len(tables))()
        for i, (t, b, rows) in enumerate(zip(tables, bases, row_lists)):
            self._check(t)
            assert t.dtype == torch.float32 and t.is_contiguous() and b.is_contiguous() and b.shape == t.shape
            arr[i].table, arr[i].base = t.data_ptr(), b.data_ptr()
            arr[i].rows = None if rows is None else rows.data_ptr()
            arr[i].n_rows = int(t.shape[0]) if rows is None else int(rows.numel())
        return arr

    def tsum_pack(self, arr, stride, wire, stream):
        _lib.check(self.lib.n2v_tsum_pack(arr, len(arr), stride, wire.data_ptr(), self._bf16(wire), stream))

    def tsum_apply(self, arr, stride, wire, stream):
        _lib.check(self.lib.n2v_tsum_apply(arr, len(arr), stride, wire.data_ptr(), self._bf16(wire), stream))

    def flush(self, x, xs, base, w, hot_pos, sum_last):
        self._check(x)
        with torch.cuda.device(x.device):
            _lib.check(self.lib.n2v_merge_flush(
                _lib.ptr(x), _lib.ptr(xs), _lib.ptr(base), int(x.shape[0]), int(x.shape[1]), _lib.ptr(w),
                _lib.ptr(hot_pos), _lib.ptr(sum_last), 0 if sum_last is None else self._bf16(sum_last),
                _lib.stream_ptr(x.device)))


class ReplicaMerger:
    """One rank's side of the merges of `tables` (fp32 [N, stride] each, trained in place).

    end_interval() = snapshot() -> all-reduce of the hot rows' changes -> finish(), pipelined over row ranges so
    that packing and folding run under the wire time; finish() also starts the all-reduce of the cold rows'
    changes and returns — that sum is folded in by the NEXT snapshot().  flush() ends the run: every rank then
    holds the same tables.  `overlap=False`: one range, and the cold all-reduce is waited for at once — same
    arithmetic, same results, nothing hidden (the A/B of the overlap).
    The simulated-replica driver calls the three phases itself."""

    def __init__(self, tables, plan, comm, overlap=True, ops=None, pipe=8, pipe_bytes=32 << 20):
        self.t, self.plan, self.comm, self.overlap = list(tables), plan, comm, bool(overlap)
        self.pipe = int(pipe) if overlap else 1
        self.pipe_bytes = int(pipe_bytes)       # a row range of the pipelined merge is at least this large
        self._bounds = [None] * len(self.t)
        self.ops = ops if ops is not None else HipMergeOps()
        dev = self.t[0].device
        wire = getattr(comm, "wire_dtype", None) or torch.float32
        self.xs = [t.clone() for t in self.t]
        self.base = [t.clone() for t in self.t]
        stride = int(self.t[0].shape[1])
        assert all(int(t.shape[1]) == stride for t in self.t)
        n_hot = sum(plan.n_hot)
        # one buffer per collective: the tables' hot rows back to back, the tables' cold wires back to back
        self.hot_wire = torch.zeros((n_hot, stride), dtype=wire, device=dev) if n_hot else None
        self.hot_views, o = [], 0
        for h in plan.n_hot:
            self.hot_views.append(self.hot_wire[o:o + h] if h else None)
            o += h
        self.has_cold = sum(plan.n_cold) > 0
        rows = [int(t.shape[0]) for t in self.t]
        self.cold_wire = [torch.zeros((sum(rows), stride), dtype=wire, device=dev) for _ in range(2)] if self.has_cold else None
        self.cold_views = None
        if self.has_cold:
            self.cold_views = []
            for buf in self.cold_wire:
                v, o = [], 0
                for r in rows:
                    v.append(buf[o:o + r])
                    o += r
                self.cold_views.append(v)
        self.cur = 0
        self.pending = None          # (handle, buffer index) of the cold all-reduce in flight
        self.n_merges = 0
        self._ev = []                # (kind, start event, end event) on the compute stream
        self._timed = dev.type == "cuda"

    # -- timing (bench.py: merge_seconds / overlap_fraction)
    def _mark(self):
        if not self._timed:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _span(self, kind, start):
        if self._timed:
            self._ev.append((kind, start, self._mark()))

    def seconds(self):
        """{'merge': compute-stream seconds inside the merge phases, 'wait': of which waiting for cold sums}."""
        if self._timed:
            torch.cuda.synchronize(self.t[0].device)
        out = {"merge": 0.0, "wait": 0.0}
        for kind, a, b in self._ev:
            out[kind] += a.elapsed_time(b) / 1e3
        out["merge"] += out["wait"]
        return out

    def release(self):
        """Drop the buffers (the timers stay readable)."""
        self.xs = self.base = self.hot_wire = self.hot_views = self.cold_wire = self.cold_views = None

    def probe_buffer(self):
        """The wire buffer that carries the bulk of the rows (bench.py times one stand-alone all-reduce of it)."""
        if self.hot_wire is not None and (not self.has_cold or sum(self.plan.n_hot) >= sum(self.plan.n_cold)):
            return self.hot_wire
        return self.cold_wire[0]

    def _wait_pending(self):
        if self.pending is None:
            return None
        handle, idx = self.pending
        t0 = self._mark()
        if handle is not None:
            handle.wait()
        self._span("wait", t0)
        self.pending = None
        return self.cold_views[idx]

    # -- phases
    def snapshot(self):
        prev = self._wait_pending()
        t0 = self._mark()
        for i, t in enumerate(self.t):
            self.ops.snapshot(t, self.xs[i], self.base[i], self.plan.w[i],
                              self.plan.hot_pos[i] if self.plan.n_hot[i] else None,
                              None if prev is None else prev[i],
                              self.cold_views[self.cur][i] if self.has_cold else None, self.hot_views[i])
        self._span("merge", t0)

    def finish(self):
        t0 = self._mark()
        for i, t in enumerate(self.t):
            if self.plan.n_hot[i]:
                self.ops.hot_apply(t, self.xs[i], self.base[i], self.plan.w[i], self.plan.hot_rows[i], self.hot_views[i])
        self._span("merge", t0)
        self._launch_cold()
        self.n_merges += 1

    def _launch_cold(self):
        if self.has_cold:
            handle = self.comm.all_reduce_async(self.cold_wire[self.cur])
            self.pending = (handle, self.cur)
            self.cur ^= 1
            if not self.overlap:
                t0 = self._mark()
                if handle is not None:
                    handle.wait()
                self._span("wait", t0)
                self.pending = (None, self.pending[1])

    def flush(self):
        last = self._wait_pending()
        t0 = self._mark()
        for i, t in enumerate(self.t):
            self.ops.flush(t, self.xs[i], self.base[i], self.plan.w[i],
                           self.plan.hot_pos[i] if self.plan.n_hot[i] else None, None if last is None else last[i])
        self._span("merge", t0)

    def _pipe_bounds(self, i):
        """Row ranges [a, b) of table i and the hot-wire positions [pa, pb) they own (hot rows ascend, so a row
        range owns a contiguous piece of the hot wire): the units of the pipelined synchronous merge."""
        if self._bounds[i] is None:
            n = int(self.t[i].shape[0])
            k = max(1, min(self.pipe, n * int(self.t[i].shape[1]) * 4 // self.pipe_bytes))
            rows = [shard_bounds(n, k, c) for c in range(k)]
            cuts = torch.tensor([a for a, _ in rows] + [n], dtype=torch.int64, device=self.plan.hot_rows[i].device)
            pos = torch.searchsorted(self.plan.hot_rows[i], cuts).tolist() if self.plan.n_hot[i] else [0] * (k + 1)
            self._bounds[i] = [(a, b, pos[c], pos[c + 1]) for c, (a, b) in enumerate(rows)]
        return self._bounds[i]

    def end_interval(self, last=False):
        """snapshot -> all-reduce of the hot rows' changes -> fold in, PIPELINED over row ranges: while range c's
        changes travel, range c+1 is being packed and range c-1 folded in, so the merge kernels run under the wire
        time (the results are those of snapshot() / finish(), bit for bit: same kernels on the same rows)."""
        prev = self._wait_pending()
        inflight = []
        for i, t in enumerate(self.t):
            hp = self.plan.hot_pos[i] if self.plan.n_hot[i] else None
            for (a, b, pa, pb) in self._pipe_bounds(i):
                t0 = self._mark()
                self.ops.snapshot(t[a:b], self.xs[i][a:b], self.base[i][a:b], self.plan.w[i][a:b],
                                  None if hp is None else hp[a:b], None if prev is None else prev[i][a:b],
                                  self.cold_views[self.cur][i][a:b] if self.has_cold else None, self.hot_views[i])
                self._span("merge", t0)
                if pb > pa:
                    inflight.append((i, pa, pb, self.comm.all_reduce_async(self.hot_views[i][pa:pb])))
        for (i, pa, pb, handle) in inflight:
            t0 = self._mark()
            if handle is not None:
                handle.wait()
            self._span("wait", t0)
            t0 = self._mark()
            self.ops.hot_apply(self.t[i], self.xs[i], self.base[i], self.plan.w[i], self.plan.hot_rows[i][pa:pb],
                               self.hot_views[i][pa:pb])
            self._span("merge", t0)
        self._launch_cold()
        self.n_merges += 1
        if last:
            self.flush()


# ---- tiered pure-sum merges (merge="tsum"): the scheme that meets the AUC band at every graph size in simulation
TSUM_STALENESS_BUDGET = 24.0   # base cadence: every row is merged once per this many tokens per row from the others
TSUM_THETA = 125.0             # expected updates by the other replicas between two merges of a row, at most
                               # (tests/probes/tsum_probe.py: 8 replicas, hub graph -0.0005 at 125, -0.0029 at 500)
TSUM_TIERS = 4                 # tier j is merged TSUM_RATIO^j times per base interval (1, 4, 16, 64)
TSUM_RATIO = 4


class SumTierPlan:
    """Per-row merge cadences for pure sums: a row whose expected updates by the other replicas per BASE interval
    lie in (theta * ratio^(j-1), theta * ratio^j] is in tier j and merged ratio^j times per base interval, so that no
    row collects more than ~theta foreign updates between two of its merges (rows beyond the last tier: as often as
    that tier).  Identical on every rank."""

    def __init__(self, counts, interval_tokens_global, world, window, negative, device, theta=TSUM_THETA,
                 n_tiers=TSUM_TIERS, ratio=TSUM_RATIO):
        self.n_tiers, self.ratio, self.world = int(n_tiers), int(ratio), world
        self.tier = []
        for upd in expected_updates(counts, interval_tokens_global, window, negative, device):
            u = (world - 1) / world * upd
            t = torch.ceil(torch.log(u.clamp_min(1e-30) / theta) / np.log(self.ratio)).clamp(0, self.n_tiers - 1).long()
            self.tier.append(torch.where(u > theta, t.clamp_min(1), torch.zeros_like(t)))
        # tiers nobody is in are dropped (a graph without hubs: one tier, no sub-intervals, full-size launches)
        self.n_tiers = 1 + max(int(t.max()) if t.numel() else 0 for t in self.tier)
        self.sub = self.ratio ** (self.n_tiers - 1)          # sub-intervals per base interval
        self.rows_ge = [[torch.arange(int(t.numel()), dtype=torch.int64, device=device)] +
                        [torch.nonzero(t >= j).flatten().contiguous() for j in range(1, self.n_tiers)] for t in self.tier]

    def level_due(self, sub_index):
        """The coarsest tier level whose merge is due after sub-interval `sub_index` (0-based, global), or None:
        level j is due every sub / ratio^j sub-intervals; a merge of level j covers every tier >= j."""
        for j in range(self.n_tiers):
            if (sub_index + 1) % (self.sub // self.ratio ** j) == 0:
                return j
        return None


class TieredSumMerger:
    """One rank's side of the tiered pure-sum merges of `tables`: merge(level) packs the changes of the rows of
    tiers >= level since THEIR last merge, all-reduces them (sum) and folds the sum in — every change is applied exactly
    once with weight 1 on every rank; only the time at which the other ranks see it depends on the row's tier."""

    def __init__(self, tables, plan, comm, ops=None, timed=False):
        self.t, self.plan, self.comm = list(tables), plan, comm
        self.ops = ops if ops is not None else HipMergeOps()
        dev = self.t[0].device
        wire = getattr(comm, "wire_dtype", None) or torch.float32
        self.base = [t.clone() for t in self.t]
        stride = int(self.t[0].shape[1])
        # ONE wire buffer for all tables (their due rows back to back): one collective per merge
        self.wire = torch.zeros((sum(int(t.shape[0]) for t in self.t), stride), dtype=wire, device=dev)
        self.n_merges = [0] * plan.n_tiers
        self._ev = []
        # timers are opt-in (bench.py): four events per merge, ~15 000 merges per pass at 8 GPUs
        self._timed = bool(timed) and dev.type == "cuda"
        # all tables of a level in one launch per step (n2v_tsum_pack / n2v_tsum_apply), arguments prepared once:
        # the hub tiers' merges are launch- and host-bound
        self.fused = hasattr(self.ops, "tsum_pack") and dev.type == "cuda"
        # the per-table path reuses the weighted merges' fold-in kernel with weight 1 on every row
        self.ones = None if self.fused else [torch.ones(int(t.shape[0]), dtype=torch.float32, device=dev) for t in self.t]
        if self.fused:
            self._stride = stride
            self._args, self._flat = [], []
            for level in range(plan.n_tiers):
                lists = [None if level == 0 else plan.rows_ge[i][level] for i in range(len(self.t))]
                self._args.append(self.ops.tsum_tables(self.t, self.base, lists))
                self._flat.append(self.wire[:sum(int(plan.rows_ge[i][level].numel()) for i in range(len(self.t)))])

    def _mark(self):
        if not self._timed:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def seconds(self):
        """{'merge': compute-stream seconds inside the merges, 'wait': of which waiting for the all-reduce}."""
        if self._timed:
            torch.cuda.synchronize(self.t[0].device)
        out = {"merge": 0.0, "wait": 0.0}
        for kind, a, b in self._ev:
            out[kind] += a.elapsed_time(b) / 1e3
        out["merge"] += out["wait"]
        return out

    def release(self):
        self.base = self.ones = self._args = self._flat = None

    def probe_buffer(self):
        return self.wire

    def pack(self, level):
        """-> (wire slice holding this rank's changes of the rows of tiers >= level of every table, per-table views)."""
        if self.fused:
            flat = self._flat[level]
            if flat.numel():
                with torch.cuda.device(flat.device):
                    self.ops.tsum_pack(self._args[level], self._stride, flat, _lib.stream_ptr(flat.device))
            return flat, None
        views, o = [], 0
        for i, t in enumerate(self.t):
            rows = self.plan.rows_ge[i][level]
            v = self.wire[o:o + int(rows.numel())]
            if rows.numel():
                self.ops.pack_rows(t, self.base[i], rows, v)
            views.append(v)
            o += int(rows.numel())
        return self.wire[:o], views

    def apply(self, level, views):
        """Folds the all-reduced wire slice of pack(level) into the tables."""
        self.n_merges[level] += 1
        if self.fused:
            flat = self._flat[level]
            if flat.numel():
                with torch.cuda.device(flat.device):
                    self.ops.tsum_apply(self._args[level], self._stride, flat, _lib.stream_ptr(flat.device))
            return
        for i, t in enumerate(self.t):
            rows = self.plan.rows_ge[i][level]
            if rows.numel():
                self.ops.hot_apply(t, t, self.base[i], self.ones[i], rows, views[i])

    def merge(self, level):
        t0 = self._mark()
        flat, views = self.pack(level)
        t1 = self._mark()
        if flat.numel():
            handle = self.comm.all_reduce_async(flat)
            if handle is not None:
                handle.wait()
        t2 = self._mark()
        self.apply(level, views)
        if self._timed:
            self._ev += [("merge", t0, t1), ("wait", t1, t2), ("merge", t2, self._mark())]


def auto_syncs(tokens_global, n_words, world):
    """Merges per pass so that (world-1) * tokens per row per interval <= the staleness budget of merge_constants."""
    if world <= 1:
        return 1
    return max(1, int(np.ceil(tokens_global * (world - 1) / (merge_constants(world)[1] * max(n_words, 1)))))


def chunk_plan(n_local, n_chunks, exact=False):
    """[begin, end) of every merge interval of a pass over n_local sentences.  exact=True keeps exactly
    n_chunks intervals (some may be empty): every rank must run the same number of collectives even when the
    shards differ in size."""
    n_chunks = max(1, int(n_chunks))
    if not exact:
        n_chunks = min(n_chunks, max(n_local, 1))
    # evenly spread (interval c = [c*n/k, (c+1)*n/k)): with more intervals than sentences the sentences must not all
    # sit in the first intervals, or the merges after them would have nothing left to merge
    return [(c * n_local // n_chunks, (c + 1) * n_local // n_chunks) for c in range(n_chunks)]
