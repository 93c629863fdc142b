"""Host side of the skip-gram / negative-sampling trainer (the gensim ``Word2Vec`` call of
src/main.py:82-90) on MI355X: vocabulary statistics and the schedule are prepared here,
every row update runs in the HIP kernel ``n2v_sgns_train`` (csrc/n2v_sgns.hip).

gensim 3.2.0 (requirements.txt:17) is a third-party dependency that is not part of the
reference tree; the statistics below restate its public ``scale_vocab`` / ``make_cum_table``
/ job-wise learning-rate decay with the arguments the reference passes
(size=d, window, min_count=0, sg=1, iter) and gensim's defaults for the rest
(negative=5, alpha=0.025, min_alpha=1e-4, sample=1e-3, ns exponent 0.75).

Multi-GPU: every rank trains a full replica on its shard of the walks
(shard = contiguous start positions, as for the walk itself) and the replicas are merged
over RCCL inside the pass and at its end (the schemes: n2v_hip/merge.py; `train` below
drives them); the learning-rate schedule is driven by the GLOBAL sentence count.
"""
import contextlib

import numpy as np
import torch

from . import _lib

UPDATE_MODES = {"plain": 0, "agent": 1, "atomic": 2}  # N2V_SGNS_* of include/n2v_hip.h
# update_mode="auto": lossless memory-side atomics unless the vocabulary has at least AUTO_AGENT_MIN_WORDS rows AND
# the corpus holds at least AUTO_AGENT_MIN_TOKENS_PER_WORD tokens per row; then agent-scope load/store (1.6-2.2x the
# pair rate).  Measured (link-prediction AUC on hub-heavy community graphs): 10 walks of 80 per node (800 tokens per
# row) — 200k nodes atomic 0.87951 / agent 0.87948; 1M nodes 0.87918 / 0.88020; 131 072 nodes atomic 0.87593 / agent
# 0.87621 against 0.87752 of the 12-thread CPU comparator: the two modes are indistinguishable and inside the +-0.002
# band.  But the agent mode's lost updates (a read-modify-write of a row can lose a concurrent one) hit hardest while
# the vectors are still growing: on the same 131 072-node graph with the main_link defaults of src/settings.py, 5 walks
# of 40 (200 tokens per row), agent trails the comparator by 0.005 (0.84200 vs 0.84716; atomic 0.84666), and after
# 2 walks of 40 by 0.24 (0.547 vs 0.787; atomic 0.784) — tests/probes/agent_band_probe.py,
# profiles/r02/logs/agent_band_probe.log.  At 3000 nodes the agent mode drifts by +0.002 and more whatever the corpus
# (every row is hot).  So short corpora and small tables keep the atomics.
AUTO_AGENT_MIN_WORDS = 1 << 17
AUTO_AGENT_MIN_TOKENS_PER_WORD = 600
# merge="hot" (per-row-weighted sums) holds the +-0.002 band on the 3 000- and 20 000-node test graphs and misses it by
# 0.0047 / 0.0064 at 4 / 8 replicas on a 131 072-node hub graph: above this vocabulary size it has to be
# asked for with allow_out_of_band=True
HOT_MERGE_MAX_WORDS = 1 << 15
# walk_splits="auto" (several wavefronts per sentence, for the short launches between hub-tier merges) only on tables of
# at least this many rows: a short launch starts all splits of a sentence at the same instant, which adds a POSITIVE AUC
# offset that grows with the number of launches and shrinks with the table size — 20k-row hub graph, tiered merges:
# +0.0026 / +0.0013 at 2 / 8 replicas with splits against +0.0014 / -0.0001 without; 131k rows with splits: +0.0010 /
# +0.0001 (tests/probes/split_bias_probe.py, profiles/r03/logs/split_bias_probe_hub20k.log)
AUTO_SPLITS_MIN_WORDS = 1 << 16
MAX_WORDS_IN_BATCH = 10000  # gensim: words per job; alpha is stepped once per job


class OutOfBandError(ValueError):
    """A mode that is known (measured) to leave BASELINE.json's +-0.002 AUC band was requested without
    allow_out_of_band=True."""
LUT_BITS = 20


def vocab_tables(counts, sample=1e-3, ns_exponent=0.75):
    """counts int64[N] (occurrences of each dense id in the corpus; 0 = not in the vocabulary)
    -> (sample_int uint32[N] or None, cum_table uint32[N]).

    gensim scale_vocab: threshold = sample * total; keep probability
    (sqrt(v/threshold) + 1) * (threshold/v) capped at 1, stored as round(p * 2^32).
    gensim make_cum_table: cum_table[i] = round(sum_{k<=i} count_k^0.75 / Z * (2^31 - 1)).
    """
    counts = np.asarray(counts, dtype=np.int64)
    v = counts.astype(np.float64)
    total = float(counts.sum())
    if total <= 0:
        raise ValueError("empty corpus")
    if not sample:
        sample_int = None
    else:
        thr = sample * total if sample < 1.0 else float(int(sample * (3 + np.sqrt(5)) / 2))
        with np.errstate(divide="ignore", invalid="ignore"):
            wp = (np.sqrt(v / thr) + 1.0) * (thr / v)
        wp = np.where(counts > 0, np.minimum(wp, 1.0), 1.0)
        sample_int = np.minimum(np.round(wp * 2.0**32), 2.0**32 - 1).astype(np.uint32)
    powc = v ** ns_exponent
    cum = np.cumsum(powc)
    z = cum[-1]
    domain = 2**31 - 1
    cum_table = np.round(cum / z * domain).astype(np.int64)
    last_nz = int(np.nonzero(counts)[0][-1])
    cum_table[last_nz:] = domain  # gensim asserts cum_table[-1] == domain
    return sample_int, cum_table.astype(np.uint32)


def check_share_negatives(share_negatives, allow_out_of_band=False):
    if share_negatives and not allow_out_of_band:
        raise OutOfBandError("share_negatives draws the negatives once per centre word — a different (correlated) sampling "
                             "scheme, +0.0025...+0.0028 AUC away from per-pair sampling at 200k / 1M nodes; pass "
                             "allow_out_of_band=True to use it")


def resolve_update_mode(requested, n_words, n_tokens, allow_out_of_band=False):
    """'auto' -> 'agent' for >= AUTO_AGENT_MIN_WORDS rows AND >= AUTO_AGENT_MIN_TOKENS_PER_WORD tokens per row, else
    'atomic'; an explicit lossy mode ('agent', 'plain') outside that region is refused unless allow_out_of_band."""
    in_region = n_words >= AUTO_AGENT_MIN_WORDS and float(n_tokens) >= AUTO_AGENT_MIN_TOKENS_PER_WORD * n_words
    if requested == "auto":
        return "agent" if in_region else "atomic"
    if requested not in UPDATE_MODES:
        raise ValueError("update_mode must be 'auto', 'atomic', 'agent' or 'plain'")
    if requested != "atomic" and not in_region and not allow_out_of_band:
        raise OutOfBandError(
            "update_mode=%r loses concurrent updates of a row; it stays inside the +-0.002 AUC band only for >= %d "
            "vocabulary rows AND >= %d corpus tokens per row (here %d rows, %.0f tokens per row: measured 0.547 vs 0.787 "
            "on a short corpus).  Use 'auto' / 'atomic', or pass allow_out_of_band=True"
            % (requested, AUTO_AGENT_MIN_WORDS, AUTO_AGENT_MIN_TOKENS_PER_WORD, n_words, float(n_tokens) / max(n_words, 1)))
    return requested


def auto_splits(n_words, n_walks, walk_length):
    """Wavefronts per sentence of a launch of n_walks walks: as many as it takes to put ~8 192 wavefronts on the chip,
    at most one per token — and none on tables below AUTO_SPLITS_MIN_WORDS rows (see there)."""
    if n_words < AUTO_SPLITS_MIN_WORDS:
        return 1
    return max(1, min(int(walk_length), -(-8192 // max(int(n_walks), 1))))


def launch_update_mode(mode_bits, auto, splits, allow_out_of_band=False):
    """update_mode word of ONE launch.  Several wavefronts on one sentence (walk_splits > 1) update the same context
    rows at the same instant: only the lossless mode was scored against the comparator for that (the C-ABI refuses the
    others) — 'auto' falls back to atomic for such launches, an explicit lossy mode needs allow_out_of_band."""
    if int(splits) > 1 and (mode_bits & 3) != UPDATE_MODES["atomic"]:
        if auto:
            return UPDATE_MODES["atomic"] | (mode_bits & 4)
        if allow_out_of_band:
            return mode_bits | 8      # N2V_SGNS_UNCHECKED
        raise OutOfBandError("walk_splits > 1 needs update_mode 'atomic' or 'auto' (or allow_out_of_band=True)")
    return mode_bits


def _row_stride(dim):
    for s in (64, 128, 256, 512):
        if dim <= s:
            return s
    raise ValueError("dimensions > 512 are not supported by the wave-per-pair kernel")


class SgnsModel:
    """Embedding tables + vocabulary statistics of one training run, on one device."""

    def __init__(self, n_words, dim=128, window=10, negative=5, alpha=0.025, min_alpha=1e-4, sample=1e-3,
                 seed=1, device=None, update_mode="auto", share_negatives=False, allow_out_of_band=False):
        if update_mode not in ("auto",) + tuple(UPDATE_MODES):
            raise ValueError("update_mode must be 'auto', 'atomic', 'agent' or 'plain'")
        check_share_negatives(share_negatives, allow_out_of_band)
        if not torch.cuda.is_available():
            raise RuntimeError("n2v_hip: no GPU visible; the SGNS trainer has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.n_words, self.dim = int(n_words), int(dim)
        self.stride = _row_stride(self.dim)
        self.window, self.negative = int(window), int(negative)
        self.alpha, self.min_alpha, self.sample, self.seed = float(alpha), float(min_alpha), sample, int(seed)
        self._auto_mode = update_mode == "auto"     # resolved by build_vocab, which knows the corpus size
        self.allow_out_of_band = bool(allow_out_of_band)
        self._share = 4 if share_negatives else 0   # N2V_SGNS_SHARE_NEGATIVES
        self._set_mode("atomic" if self._auto_mode else update_mode)
        d = self.device
        self.syn0 = torch.empty((self.n_words, self.stride), dtype=torch.float32, device=d)
        self.syn1neg = torch.empty((self.n_words, self.stride), dtype=torch.float32, device=d)
        self.pair_count = torch.zeros(1, dtype=torch.int64, device=d)
        # in-order hand-out of the sentences to the wavefronts (n2v_sgns_train's work_counter); None: static grid stride
        self.work_counter = torch.zeros(1, dtype=torch.int64, device=d)
        self.counts = None
        self.sample_int = self.cum_table = self.lut = None
        self.reset_weights()

    def _set_mode(self, name):
        self.update_mode_name = name
        self.update_mode = UPDATE_MODES[name] | self._share

    def _stream(self):
        return _lib.stream_ptr(self.device)

    def reset_weights(self):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.n2v_sgns_init(_lib.ptr(self.syn0), _lib.ptr(self.syn1neg), self.n_words, self.dim,
                                              self.stride, self.seed & (2**64 - 1), self._stream()))

    def build_vocab(self, walks=None, counts=None):
        """Word counts from a device corpus (int32 [W, L], -1 padded) or given directly."""
        d = self.device
        if counts is None:
            flat = walks.reshape(-1)
            flat = flat[flat >= 0].long()
            counts_t = torch.bincount(flat, minlength=self.n_words)
        else:
            counts_t = torch.as_tensor(counts, dtype=torch.int64, device=d)
        self.counts = counts_t.cpu().numpy()
        self._set_mode(resolve_update_mode("auto" if self._auto_mode else self.update_mode_name, self.n_words,
                                           float(self.counts.sum()), self.allow_out_of_band))
        sample_int, cum = vocab_tables(self.counts, self.sample)
        self.sample_int = None if sample_int is None else torch.from_numpy(sample_int.view(np.int32)).to(d)
        self.cum_table = torch.from_numpy(cum.view(np.int32)).to(d)
        self.lut = torch.empty((1 << LUT_BITS) + 1, dtype=torch.int32, device=d)
        with torch.cuda.device(d):
            _lib.check(self.lib.n2v_build_neg_lut(_lib.ptr(self.cum_table), self.n_words, LUT_BITS,
                                                  _lib.ptr(self.lut), self._stream()))

    def train_pass(self, walks, lens, sentences_base, sentences_total, walk_id_base, sentences_step=1,
                   max_blocks=0, splits=1):
        """One kernel launch over `walks` (device int32 [n, L]); asynchronous.  splits: wavefronts per walk ("auto":
        as many as it takes to put ~8 192 wavefronts on the chip, for the short launches of the tiered merges)."""
        assert walks.dtype == torch.int32 and walks.is_contiguous() and walks.device == self.device
        n, L = int(walks.shape[0]), int(walks.shape[1])
        if n == 0:
            return
        with torch.cuda.device(self.device):
            self._launch(_lib.ptr(walks), _lib.ptr(lens), n, L, sentences_base, sentences_step, sentences_total,
                         walk_id_base, max_blocks, splits, self._stream())

    def _launch(self, walks_ptr, lens_ptr, n, L, sentences_base, sentences_step, sentences_total, walk_id_base,
                max_blocks, splits, stream):
        if splits == "auto":
            splits = auto_splits(self.n_words, n, L)
        mode = launch_update_mode(self.update_mode, self._auto_mode, splits, self.allow_out_of_band)
        _lib.check(self.lib.n2v_sgns_train(
            walks_ptr, lens_ptr, n, L, _lib.ptr(self.syn0), _lib.ptr(self.syn1neg), self.n_words,
            self.dim, self.stride, self.window, self.negative, _lib.ptr(self.sample_int),
            _lib.ptr(self.cum_table), _lib.ptr(self.lut), LUT_BITS, self.alpha, self.min_alpha,
            int(sentences_base), int(sentences_step), int(sentences_total), max(1, MAX_WORDS_IN_BATCH // L),
            self.seed & (2**64 - 1),
            int(walk_id_base), _lib.ptr(self.pair_count), mode, int(max_blocks), int(splits),
            _lib.ptr(self.work_counter), stream))

    def span_trainer(self, walks, lens, sentences_total, sentences_step, splits="auto"):
        """-> launch(b, e, sentences_base, walk_id_base): train_pass over walks[b:e] without building tensor views —
        for drivers that issue thousands of short launches per pass (the tiered merges).  The caller holds
        torch.cuda.device(self.device) and keeps `walks` / `lens` alive."""
        assert walks.dtype == torch.int32 and walks.is_contiguous() and walks.device == self.device
        L = int(walks.shape[1])
        wp = walks.data_ptr()
        lp = None if lens is None else lens.data_ptr()
        if lens is not None:
            assert lens.dtype == torch.int32 and lens.is_contiguous()
        dev = self.device

        def launch(b, e, sentences_base, walk_id_base):
            if e > b:
                self._launch(wp + b * L * 4, None if lp is None else lp + b * 4, e - b, L, sentences_base, sentences_step,
                             sentences_total, walk_id_base, 0, splits, _lib.stream_ptr(dev))
        return launch

    def span_launcher(self, walks, lens, sentences_total, sentences_step, interval_state, subs_per_interval, n_sub_total,
                      shard_offset, splits):
        """-> launch(sub_index): n2v_sgns_train_span over this rank's whole shard — the walk range comes from the device
        word pair `interval_state` (int64[2]: base interval index, sentences of earlier epochs), so a captured launch
        can be replayed for every base interval of a pass (the graph path of the tiered merges)."""
        assert walks.dtype == torch.int32 and walks.is_contiguous() and walks.device == self.device
        assert interval_state.dtype == torch.int64 and interval_state.numel() == 2 and interval_state.device == self.device
        n_local, L = int(walks.shape[0]), int(walks.shape[1])
        max_walks = -(-n_local // int(n_sub_total))
        if splits == "auto":
            splits = auto_splits(self.n_words, max_walks, L)
        mode = launch_update_mode(self.update_mode, self._auto_mode, splits, self.allow_out_of_band)
        dev = self.device

        def launch(sub_index):
            _lib.check(self.lib.n2v_sgns_train_span(
                _lib.ptr(walks), _lib.ptr(lens), n_local, L, _lib.ptr(self.syn0), _lib.ptr(self.syn1neg), self.n_words,
                self.dim, self.stride, self.window, self.negative, _lib.ptr(self.sample_int), _lib.ptr(self.cum_table),
                _lib.ptr(self.lut), LUT_BITS, self.alpha, self.min_alpha, int(sentences_step), int(sentences_total),
                max(1, MAX_WORDS_IN_BATCH // L), self.seed & (2**64 - 1), _lib.ptr(self.pair_count), mode, 0, int(splits),
                _lib.ptr(interval_state), int(sub_index), int(subs_per_interval), int(n_sub_total), int(shard_offset),
                _lib.ptr(self.work_counter), _lib.stream_ptr(dev)))
        return launch

    def pairs_trained(self):
        return int(self.pair_count.item())

    def vectors(self):
        """syn0 without the padding columns (device view)."""
        return self.syn0[:, :self.dim]


class _ProcessGroupComm:
    """torch.distributed all-reduce (RCCL on GPUs, gloo in the CPU tests)."""
    wire_dtype = None   # dtype the replicas' changes travel in (None: float32)

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.graph_capturable = str(dist.get_backend(group)) == "nccl"     # RCCL collectives can be captured into a HIP graph

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)

    def all_reduce_async(self, t):
        """Starts the all-reduce of `t` (in place) and returns a handle; handle.wait() orders everything issued
        afterwards on the current stream behind its completion.  Over RCCL the collective runs on the process
        group's own stream, i.e. under whatever the caller launches before it waits."""
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=True)


# the merge schemes live in n2v_hip/merge.py; their names stay importable from here (tests, probes, bench.py)
from .merge import (  # noqa: F401
    HOT_BUDGET, HOT_THETA, STALENESS_BUDGET, MERGE_CONSTANTS, merge_constants, MIN_WALKS_PER_LAUNCH,
    expected_updates, merge_weights, MergePlan, HipMergeOps, ReplicaMerger, TSUM_STALENESS_BUDGET, TSUM_THETA,
    TSUM_TIERS, TSUM_RATIO, SumTierPlan, TieredSumMerger, auto_syncs, chunk_plan, shard_bounds)


def _merge_setup(model, L, n_walks_global, world, syncs_per_epoch, merge, cold_delay):
    """(n_chunks, MergePlan) — everything that decides how many collectives run is derived from GLOBAL quantities,
    never from a rank's shard size."""
    n_chunks = (auto_syncs(n_walks_global * L, model.n_words, world) if syncs_per_epoch == "auto"
                else int(syncs_per_epoch))
    n_chunks = max(1, min(n_chunks, max(1, n_walks_global // world)))
    plan = MergePlan(model.counts, n_walks_global * L / n_chunks, world, model.window, model.negative, model.device,
                     mode=merge, cold_delay=cold_delay)
    return n_chunks, plan


def _tsum_setup(model, L, n_walks_global, world, syncs_per_epoch):
    n_chunks = (max(1, int(np.ceil(n_walks_global * L * (world - 1) / (TSUM_STALENESS_BUDGET * max(model.n_words, 1)))))
                if syncs_per_epoch == "auto" else int(syncs_per_epoch))
    n_chunks = max(1, min(n_chunks, max(1, n_walks_global // world)))
    plan = SumTierPlan(model.counts, n_walks_global * L / n_chunks, world, model.window, model.negative, model.device)
    return n_chunks, plan


def _train_tsum(model, walks, lens, epochs, comm, n_walks_global, shard_offset, syncs_per_epoch, ops, timers=False,
                graph="auto", splits="auto"):
    """merge="tsum": pure sums at per-row cadences (SumTierPlan).  The pass is cut into base intervals x sub-intervals;
    after every sub-interval the tiers that are due are merged.  Meets the AUC band at every graph size in simulation
    (DESIGN.md 6), but the launches between two hub-tier merges are short (n_local / (n_chunks * 64) walks): the
    price of synchronous hub tiers with one wavefront per walk.  On a GPU the [train, pack, all-reduce, apply] x sub
    sequence of ONE base interval is captured into a HIP graph (the merge pattern repeats every base interval; the
    training launches read their walk range from a device counter) and replayed n_chunks times per pass — ~15 000
    short launches per pass at 8 GPUs then cost no host time (graph=False / timers=True: the eager loop)."""
    n_local = int(walks.shape[0])
    world = comm.world
    total = epochs * n_walks_global
    n_chunks, plan = _tsum_setup(model, int(walks.shape[1]), n_walks_global, world, syncs_per_epoch)
    merger = TieredSumMerger([model.syn0, model.syn1neg], plan, comm, ops=ops, timed=timers)
    use_graph = (graph is not False and not timers and model.device.type == "cuda" and hasattr(model, "span_launcher")
                 and getattr(comm, "graph_capturable", True) and merger.fused)
    if use_graph:
        try:
            _train_tsum_graph(model, walks, lens, epochs, world, n_walks_global, shard_offset, n_chunks, plan, merger, total,
                              splits)
            return merger
        except _GraphCaptureFailed as e:
            if graph is True:
                raise
            import warnings
            warnings.warn("tiered merges: HIP graph capture failed (%s); running the eager loop" % (e,))
    subs = chunk_plan(n_local, n_chunks * plan.sub, exact=True)
    due = [plan.level_due(c) for c in range(len(subs))]
    launch = model.span_trainer(walks, lens, sentences_total=total, sentences_step=world, splits=splits)
    # (the host-logic tests drive this loop with a stand-in model on CPU tensors)
    with (torch.cuda.device(model.device) if model.device.type == "cuda" else contextlib.nullcontext()):
        for ep in range(epochs):
            for c, (b, e) in enumerate(subs):
                launch(b, e, ep * n_walks_global + b * world, ep * n_walks_global + shard_offset + b)
                if due[c] is not None:
                    merger.merge(due[c])
    return merger


class _GraphCaptureFailed(RuntimeError):
    pass


def _train_tsum_graph(model, walks, lens, epochs, world, n_walks_global, shard_offset, n_chunks, plan, merger, total, splits):
    """One base interval as a HIP graph, replayed n_chunks x epochs times.  Nothing is trained during capture; a failed
    capture leaves the tables untouched (the caller falls back to the eager loop)."""
    dev = model.device
    sub = plan.sub
    due = [plan.level_due(j) for j in range(sub)]            # periodic in the base interval
    with torch.cuda.device(dev):
        state = torch.zeros(2, dtype=torch.int64, device=dev)       # {base interval, sentences of earlier epochs}
        step = torch.tensor([1, 0], dtype=torch.int64, device=dev)
        launch = model.span_launcher(walks, lens, total, world, state, sub, n_chunks * sub, shard_offset, splits)
        n_before = list(merger.n_merges)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        try:
            with torch.cuda.graph(g):
                for j in range(sub):
                    launch(j)
                    if due[j] is not None:
                        merger.merge(due[j])
                state.add_(step)
        except Exception as e:          # capture is all-or-nothing: nothing has run
            merger.n_merges = n_before
            raise _GraphCaptureFailed("%s: %s" % (type(e).__name__, e))
        per_interval = [a - b for a, b in zip(merger.n_merges, n_before)]
        for ep in range(epochs):
            state.copy_(torch.tensor([0, ep * n_walks_global], dtype=torch.int64), non_blocking=False)
            for _ in range(n_chunks):
                g.replay()
        merger.n_merges = [b + k * n_chunks * epochs for b, k in zip(n_before, per_interval)]
        merger.graph_replays = n_chunks * epochs
        torch.cuda.current_stream(dev).synchronize()      # the graph and its buffers go out of scope with this frame
    return merger


def check_merge_in_band(merge, n_words, allow_out_of_band=False):
    """merge="hot" above HOT_MERGE_MAX_WORDS rows is known to leave the band: refuse unless asked for explicitly."""
    if merge not in ("tsum", "hot"):
        raise ValueError("merge must be 'tsum' or 'hot'")
    if merge == "hot" and n_words > HOT_MERGE_MAX_WORDS and not allow_out_of_band:
        raise OutOfBandError(
            "merge='hot' (per-row-weighted sums) was measured 0.0047 / 0.0064 AUC below the sequential comparator at 4 / 8 "
            "replicas on a 131 072-node graph (inside the +-0.002 band only on the 3 000- and 20 000-node test graphs); "
            "%d rows > %d: use merge='tsum' or pass allow_out_of_band=True" % (n_words, HOT_MERGE_MAX_WORDS))


def train(model, walks, lens, epochs=1, comm=None, n_walks_global=None, shard_offset=0, syncs_per_epoch="auto",
          merge="tsum", overlap=True, cold_delay=False, ops=None, timers=False, graph="auto", splits="auto"):
    """Train `epochs` passes over this rank's walks.  With a communicator the replicas are merged: merge="tsum"
    (default) by pure sums at per-row cadences (TieredSumMerger) — the scheme that stays inside the AUC band at
    every graph size measured —, merge="hot" by per-row-weighted sums at `syncs_per_epoch` merges per pass
    (ReplicaMerger; several times faster at 8 GPUs, inside the band on the small probe graphs, 0.003-0.006 off at
    131k nodes).  Returns the merger (None on one GPU); timers=True makes the tiered-sum merger record the stream
    time of its merges (`seconds()`; the weighted merger always does — it merges a hundred times per pass, not 15 000)."""
    n_local = int(walks.shape[0])
    if n_walks_global is None:
        n_walks_global = n_local
    total = epochs * n_walks_global
    world = comm.world if comm is not None else 1
    if world == 1:
        for ep in range(epochs):
            model.train_pass(walks, lens, sentences_base=ep * n_walks_global, sentences_step=1,
                             sentences_total=total, walk_id_base=ep * n_walks_global + shard_offset)
        return None
    check_merge_in_band(merge, model.n_words, getattr(model, "allow_out_of_band", False))
    if merge == "tsum":
        return _train_tsum(model, walks, lens, epochs, comm, n_walks_global, shard_offset, syncs_per_epoch, ops, timers,
                           graph=graph, splits=splits)
    n_chunks, plan = _merge_setup(model, int(walks.shape[1]), n_walks_global, world, syncs_per_epoch, merge, cold_delay)
    merger = ReplicaMerger([model.syn0, model.syn1neg], plan, comm, overlap=overlap, ops=ops)
    chunks = chunk_plan(n_local, n_chunks, exact=True)
    for ep in range(epochs):
        for i, (b, e) in enumerate(chunks):
            if e > b:
                # all replicas advance together: `b` local sentences = b * world global ones
                model.train_pass(walks[b:e], None if lens is None else lens[b:e],
                                 sentences_base=ep * n_walks_global + b * world, sentences_step=world,
                                 sentences_total=total, walk_id_base=ep * n_walks_global + shard_offset + b, splits="auto")
            merger.end_interval(last=(ep + 1 == epochs and i + 1 == len(chunks)))
    return merger


class _SimGroup:
    """The collectives of G replicas that live in ONE process (validation only).  Sums are formed in the wire
    dtype one replica after the other, like a ring all-reduce in that dtype."""

    def __init__(self, world, wire_dtype=None):
        self.world, self.wire_dtype = world, wire_dtype
        self._queue = []

    class _Comm:
        def __init__(self, group):
            self.group, self.world, self.wire_dtype = group, group.world, group.wire_dtype

        def all_reduce_async(self, t):
            self.group._queue.append(t)
            if len(self.group._queue) == self.world:
                _SimGroup.reduce(self.group._queue)
                self.group._queue = []
            return None

        def all_reduce_sum(self, t):
            raise RuntimeError("simulated replicas: the driver reduces the hot wires itself")

    def comm(self):
        return _SimGroup._Comm(self)

    @staticmethod
    def reduce(tensors):
        acc = tensors[0].clone()
        for t in tensors[1:]:
            acc += t
        for t in tensors:
            t.copy_(acc)


def train_simulated_replicas(models, shards, n_walks_global, syncs_per_epoch="auto", merge="tsum", epochs=1,
                             cold_delay=False, wire_dtype=torch.bfloat16, splits="auto"):
    """Validation helper: `models` are G replicas on one device, `shards[r] = (walks, lens, shard_offset)` what
    rank r would hold.  Runs the schedule of `train` with one ReplicaMerger per replica — the same kernels, the
    same two tiers, the same one-interval delay of the cold rows — so the multi-GPU scheme can be scored for AUC
    on a one-GPU box.  Returns the number of merges per pass."""
    G = len(models)
    L = int(shards[0][0].shape[1])
    if merge == "tsum":
        n_chunks, plan = _tsum_setup(models[0], L, n_walks_global, G, syncs_per_epoch)
        group = _SimGroup(G, wire_dtype)
        mergers = [TieredSumMerger([m.syn0, m.syn1neg], plan, group.comm()) for m in models]
        subs = [chunk_plan(int(w.shape[0]), n_chunks * plan.sub, exact=True) for w, _, _ in shards]
        total = epochs * n_walks_global
        for ep in range(epochs):
            for c in range(n_chunks * plan.sub):
                for r, m in enumerate(models):
                    w, l, off = shards[r]
                    b, e = subs[r][c]
                    if e > b:
                        m.train_pass(w[b:e], None if l is None else l[b:e], sentences_base=ep * n_walks_global + b * G,
                                     sentences_step=G, sentences_total=total, walk_id_base=ep * n_walks_global + off + b,
                                     splits=splits)
                level = plan.level_due(c)
                if level is None:
                    continue
                packed = [mg.pack(level) for mg in mergers]
                if packed[0][0].numel():
                    _SimGroup.reduce([flat for flat, _ in packed])
                for mg, (_, v) in zip(mergers, packed):
                    mg.apply(level, v)
        return n_chunks
    n_chunks, plan = _merge_setup(models[0], L, n_walks_global, G, syncs_per_epoch, merge, cold_delay)
    group = _SimGroup(G, wire_dtype)
    mergers = [ReplicaMerger([m.syn0, m.syn1neg], plan, group.comm()) for m in models]
    plans = [chunk_plan(int(w.shape[0]), n_chunks, exact=True) for w, _, _ in shards]
    total = epochs * n_walks_global
    for ep in range(epochs):
        for c in range(n_chunks):
            for r, m in enumerate(models):
                w, l, off = shards[r]
                b, e = plans[r][c]
                if e > b:
                    m.train_pass(w[b:e], None if l is None else l[b:e], sentences_base=ep * n_walks_global + b * G,
                                 sentences_step=G, sentences_total=total, walk_id_base=ep * n_walks_global + off + b,
                                     splits=splits)
            for mg in mergers:
                mg.snapshot()
            if mergers[0].hot_wire is not None:
                _SimGroup.reduce([mg.hot_wire for mg in mergers])
            for mg in mergers:
                mg.finish()
    for mg in mergers:
        mg.flush()
    return n_chunks


# --------------------------------------------------------------------------- gensim-like results
class _VocabEntry:
    __slots__ = ("index", "count")

    def __init__(self, index, count):
        self.index, self.count = index, count


class KeyedVectors:
    """The part of gensim's KeyedVectors the reference's consumers use
    (src/main_link.py:43-61,130,358-365): ``wv[str(id)]``, ``wv.similarity(a, b)``,
    ``wv.vocab`` (keys are str(id)), ``save_word2vec_format(path)``; words are ordered by
    descending corpus count like gensim's sorted vocabulary."""

    def __init__(self, labels, counts, vectors):
        order = np.argsort(-counts, kind="stable")
        order = order[counts[order] > 0]
        self.index2word = [str(int(labels[i])) for i in order]
        self.syn0 = np.ascontiguousarray(vectors[order], dtype=np.float32)
        self.vectors = self.syn0
        self.vocab = {w: _VocabEntry(i, int(counts[order[i]])) for i, w in enumerate(self.index2word)}
        self.vector_size = self.syn0.shape[1] if self.syn0.ndim == 2 else 0

    def __getitem__(self, word):
        if isinstance(word, (list, tuple, np.ndarray)):
            return np.vstack([self[w] for w in word])
        return self.syn0[self.vocab[word].index]

    def __contains__(self, word):
        return word in self.vocab

    def similarity(self, w1, w2):
        a, b = self[w1], self[w2]
        return float(np.dot(a / np.linalg.norm(a), b / np.linalg.norm(b)))

    def save_word2vec_format(self, fname):
        with open(fname, "w") as f:
            f.write("%d %d\n" % (len(self.index2word), self.vector_size))
            for w, row in zip(self.index2word, self.syn0):
                f.write("%s %s\n" % (w, " ".join("%f" % x for x in row)))


class Word2VecResult:
    """What ``learn_embeddings`` returns in src/main.py:82-90 (the model; ``.wv`` is what
    src/main_link.py:36-41 returns)."""

    def __init__(self, wv, model, pairs):
        self.wv = wv
        self.sgns = model
        self.pairs_trained = pairs

    def __getitem__(self, word):
        return self.wv[word]

    def save_word2vec_format(self, fname):
        self.wv.save_word2vec_format(fname)


def smoke_check(graph, corpus):
    """One tiny SGNS pass on the walks of __graft_entry__.smoke(); checks that training
    moved the tables, produced finite values and counted the pairs the windows imply."""
    csr = graph._csr
    m = SgnsModel(csr.n_nodes, dim=128, window=5, negative=5, sample=0, seed=7, device=corpus.walks.device)
    m.build_vocab(corpus.walks)
    before = m.syn0.clone()
    train(m, corpus.walks, corpus.lens, epochs=1)
    torch.cuda.synchronize()
    assert torch.isfinite(m.syn0).all() and torch.isfinite(m.syn1neg).all()
    assert (m.syn0 != before).any() and (m.syn1neg != 0).any()
    assert (m.syn0[:, m.dim:] == 0).all()
    n_pairs = m.pairs_trained()
    L = corpus.walks.shape[1]
    W = corpus.walks.shape[0]
    assert 0 < n_pairs <= W * L * 2 * 5, n_pairs
    print("smoke: SGNS pass trained %d pairs" % n_pairs)
