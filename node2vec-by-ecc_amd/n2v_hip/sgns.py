"""Host side of the skip-gram / negative-sampling trainer (the gensim ``Word2Vec`` call of
src/main.py:82-90) on MI355X: vocabulary statistics and the schedule are prepared here,
every row update runs in the HIP kernel ``n2v_sgns_train`` (csrc/n2v_sgns.hip).

gensim 3.2.0 (requirements.txt:17) is a third-party dependency that is not part of the
reference tree; the statistics below restate its public ``scale_vocab`` / ``make_cum_table``
/ job-wise learning-rate decay with the arguments the reference passes
(size=d, window, min_count=0, sg=1, iter) and gensim's defaults for the rest
(negative=5, alpha=0.025, min_alpha=1e-4, sample=1e-3, ns exponent 0.75).

Multi-GPU: every rank trains a full replica on its shard of the walks
(shard = contiguous start positions, as for the walk itself) and the two tables are
all-reduced over RCCL at sync points inside the epoch and at its end; the learning-rate
schedule is driven by the GLOBAL sentence count.
"""
import numpy as np
import torch

from . import _lib

UPDATE_MODES = {"plain": 0, "agent": 1, "atomic": 2}  # N2V_SGNS_* of include/n2v_hip.h
# update_mode="auto": lossless memory-side atomics for small vocabularies, agent-scope
# load/store above this many rows.  Measured (tools/mode_auc_probe.py, hub-heavy community graphs,
# link-prediction AUC, two seeds each): 200k nodes atomic 0.87951 / agent 0.87948; 1M nodes atomic
# 0.87918 / agent 0.88020 — inside the +-0.002 band at 1.6-2.2x the pair rate; at 3000 nodes the
# agent mode drifts by +0.002 and more (every row is hot), so small tables keep the atomics.
AUTO_AGENT_MIN_WORDS = 1 << 17
MAX_WORDS_IN_BATCH = 10000  # gensim: words per job; alpha is stepped once per job
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


def _row_stride(dim):
    for s in (64, 128, 256, 512):
        if dim <= s:
            return s
    raise ValueError("dimensions > 512 are not supported by the wave-per-pair kernel")


class SgnsModel:
    """Embedding tables + vocabulary statistics of one training run, on one device."""

    def __init__(self, n_words, dim=128, window=10, negative=5, alpha=0.025, min_alpha=1e-4, sample=1e-3,
                 seed=1, device=None, update_mode="auto", share_negatives=False):
        if not torch.cuda.is_available():
            raise RuntimeError("n2v_hip: no GPU visible; the SGNS trainer has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.n_words, self.dim = int(n_words), int(dim)
        self.stride = _row_stride(self.dim)
        self.window, self.negative = int(window), int(negative)
        self.alpha, self.min_alpha, self.sample, self.seed = float(alpha), float(min_alpha), sample, int(seed)
        if update_mode == "auto":
            update_mode = "agent" if int(n_words) >= AUTO_AGENT_MIN_WORDS else "atomic"
        self.update_mode_name = update_mode
        self.update_mode = UPDATE_MODES[update_mode] | (4 if share_negatives else 0)  # N2V_SGNS_SHARE_NEGATIVES
        d = self.device
        self.syn0 = torch.empty((self.n_words, self.stride), dtype=torch.float32, device=d)
        self.syn1neg = torch.empty((self.n_words, self.stride), dtype=torch.float32, device=d)
        self.pair_count = torch.zeros(1, dtype=torch.int64, device=d)
        self.counts = None
        self.sample_int = self.cum_table = self.lut = None
        self.reset_weights()

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
        sample_int, cum = vocab_tables(self.counts, self.sample)
        self.sample_int = None if sample_int is None else torch.from_numpy(sample_int.view(np.int32)).to(d)
        self.cum_table = torch.from_numpy(cum.view(np.int32)).to(d)
        self.lut = torch.empty((1 << LUT_BITS) + 1, dtype=torch.int32, device=d)
        with torch.cuda.device(d):
            _lib.check(self.lib.n2v_build_neg_lut(_lib.ptr(self.cum_table), self.n_words, LUT_BITS,
                                                  _lib.ptr(self.lut), self._stream()))

    def train_pass(self, walks, lens, sentences_base, sentences_total, walk_id_base, sentences_step=1,
                   max_blocks=0):
        """One kernel launch over `walks` (device int32 [n, L]); asynchronous."""
        assert walks.dtype == torch.int32 and walks.is_contiguous() and walks.device == self.device
        n, L = int(walks.shape[0]), int(walks.shape[1])
        if n == 0:
            return
        alpha_batch = max(1, MAX_WORDS_IN_BATCH // L)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.n2v_sgns_train(
                _lib.ptr(walks), _lib.ptr(lens), n, L, _lib.ptr(self.syn0), _lib.ptr(self.syn1neg), self.n_words,
                self.dim, self.stride, self.window, self.negative, _lib.ptr(self.sample_int),
                _lib.ptr(self.cum_table), _lib.ptr(self.lut), LUT_BITS, self.alpha, self.min_alpha,
                int(sentences_base), int(sentences_step), int(sentences_total), alpha_batch,
                self.seed & (2**64 - 1),
                int(walk_id_base), _lib.ptr(self.pair_count), self.update_mode, int(max_blocks), self._stream()))

    def pairs_trained(self):
        return int(self.pair_count.item())

    def vectors(self):
        """syn0 without the padding columns (device view)."""
        return self.syn0[:, :self.dim]


class _ProcessGroupComm:
    """torch.distributed all-reduce (RCCL on GPUs, gloo in the CPU tests)."""
    wire_dtype = None   # dtype the replicas' changes travel in (None: the tables themselves, fp32)

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_reduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)


def shard_bounds(n_items, world, rank):
    """Contiguous shard [begin, end) of n_items for `rank` (as src/main_link.py:261-264 splits
    start nodes among its pool workers)."""
    per = -(-n_items // world)
    b = min(rank * per, n_items)
    return b, min(b + per, n_items)


def merge_replicas(tables, bases, comm, mode="hot", weights=None):
    """Combine the replicas' tables in place at a sync point; `bases` holds the last merged copy.
    mode 'hot' (default): base + w_row * (sum of every replica's change), w_row in [1/world, 1]
                 from merge_weights(): rows that receive few updates per interval get the SUM of
                 the changes (what one shared Hogwild table would have received), rows that every
                 replica hammers (hubs, frequent negatives) get their MEAN — summing those
                 overshoots by a factor `world`, averaging cold rows under-trains them by it.
    mode 'delta': w_row = 1 (pure sum).   mode 'avg': w_row = 1/world (local SGD)."""
    wire = getattr(comm, "wire_dtype", None)
    for i, (t, b) in enumerate(zip(tables, bases)):
        if wire is not None and mode in ("hot", "delta"):
            # the replicas' CHANGES travel, as `wire` (bfloat16 over RCCL: half the bytes; AUC unchanged to 1e-4 on
            # both probe graphs, profiles/r01/logs/replica_bf16_*.log)
            d = (t - b).to(wire)
            comm.all_reduce_sum(d)
            d = d.to(t.dtype)
            if mode == "hot":
                d.mul_(weights[i][:, None])
            torch.add(b, d, out=t)
            b.copy_(t)
            continue
        comm.all_reduce_sum(t)
        if mode == "avg":
            t.div_(comm.world)
        elif mode == "delta":
            t.sub_(b, alpha=comm.world - 1)
        elif mode == "hot":
            t.sub_(b, alpha=comm.world).mul_(weights[i][:, None]).add_(b)
        else:
            raise ValueError("merge mode %r" % (mode,))
        if b is not None:
            b.copy_(t)


HOT_BUDGET = 256.0  # updates per replica and interval above which a row is merged towards the mean
HOT_EVERY = 8       # merges of the hot tier per full merge
MIN_WALKS_PER_LAUNCH = 8192  # one wavefront trains one walk at a time; launches of 5 356 walks still run at the
                             # full-pass rate (tools/sgns_grid_probe.py), much shorter ones have not been measured
HOT_TIER_FACTOR = 2.0  # a row is in the hot tier when its expected updates per full interval exceed this many budgets


def merge_weights(counts, interval_tokens_global, world, window, negative, device, budget=HOT_BUDGET, with_lam=False):
    """Per-row merge weights (w_syn0, w_syn1neg) of mode 'hot'.  Expected updates of row v per
    interval: syn0 (context rows) ~ pairs_per_token * T * p_v; syn1neg (targets) ~
    pairs_per_token * T * (p_v + negative * p_neg_v), T = tokens of all replicas per interval,
    p the unigram and p_neg the unigram^0.75 distribution.  With u = (world-1)/world * updates,
    lambda = min(1, budget / u) and w = lambda + (1 - lambda) / world."""
    c = torch.as_tensor(counts, dtype=torch.float64, device=device)
    pv = c / c.sum().clamp_min(1)
    pn = c ** 0.75
    pn = pn / pn.sum().clamp_min(1e-300)
    ppt = window + 0.5
    out, lams = [], []
    for upd in (ppt * interval_tokens_global * pv, ppt * interval_tokens_global * (pv + negative * pn)):
        u = (world - 1) / world * upd
        lam = torch.clamp(budget / u.clamp_min(1e-30), max=1.0)
        lams.append(lam)
        out.append((lam + (1 - lam) / world).to(torch.float32))
    return (out, lams) if with_lam else out


class TierPlan:
    """Two-tier merge schedule of mode 'hot'.  Rows that every replica hammers (hubs, frequent negatives: expected
    updates per full interval above HOT_TIER_FACTOR budgets) are merged `every` times per full interval — a message
    of a few per cent of the table — so their replicas never drift far apart; all rows are merged once per full
    interval.  Weights are those of merge_weights for the time a row actually waited.  Measured effect and when
    it is switched on: hot_every_for() and DESIGN.md section 6."""

    def __init__(self, counts, interval_tokens_global, world, window, negative, device, every=HOT_EVERY,
                 factor=HOT_TIER_FACTOR, budget=HOT_BUDGET):
        w_full, lam_full = merge_weights(counts, interval_tokens_global, world, window, negative, device, budget, True)
        w_sub = merge_weights(counts, interval_tokens_global / max(every, 1), world, window, negative, device, budget)
        self.every = int(every)
        self.rows, self.w_rows, self.w_full = [], [], []
        for wf, lf, ws in zip(w_full, lam_full, w_sub):
            hot = lf < 1.0 / factor
            rows = torch.nonzero(hot).flatten()
            self.rows.append(rows)
            self.w_rows.append(ws[rows].contiguous())
            self.w_full.append(torch.where(hot, ws, wf))
        if self.every <= 1 or all(r.numel() == 0 for r in self.rows):
            self.every = 1
            self.w_full = w_full

    def sub_intervals(self, n_full):
        return n_full * self.every


def merge_hot_rows(tables, bases, comm, plan):
    """The hot tier's merge: same arithmetic as merge_replicas(mode='hot') on the gathered rows only."""
    for t, b, rows, w in zip(tables, bases, plan.rows, plan.w_rows):
        if rows.numel() == 0:
            continue
        wire = getattr(comm, "wire_dtype", None)
        bb = b.index_select(0, rows)
        if wire is not None:
            x = (t.index_select(0, rows) - bb).to(wire)
            comm.all_reduce_sum(x)
            x = x.to(t.dtype).mul_(w[:, None]).add_(bb)
        else:
            x = t.index_select(0, rows)
            comm.all_reduce_sum(x)
            x.sub_(bb, alpha=comm.world).mul_(w[:, None]).add_(bb)
        t.index_copy_(0, rows, x)
        b.index_copy_(0, rows, x)


# Sync cadence.  Measured on one MI355X by training G simulated replicas
# (tests/probes/replica_auc_probe.py): on a 3000-node uniform graph (CPU comparator AUC 0.8961) the pure
# sum ('delta') stays within 0.0005 while (G-1) * tokens per vocabulary row per interval is about
# 12-22, is off by 0.0023 at 50 and diverges at 87.  On a 20k-node graph WITH hubs (comparator
# 0.8672) no cadence rescues the pure sum (+0.006 at G=2, +0.020 at G=8: hub rows overshoot) nor
# the mean (-0.014 / -0.050: cold rows under-train).  With the 'hot' interpolation the cadence can be relaxed:
# budgets 24 / 48 / 96 give +0.0013 / +0.0016 / +0.0022 (G=2) and -0.0003 / +0.0010 / +0.0026 (G=8) on the
# uniform graph and -0.0023 / -0.0015 / -0.0031 (G=2), -0.0060 / -0.0030 / -0.0039 (G=8) on the hub graph:
# 48 is the largest budget inside the +-0.002 band on the uniform graph and the best one on the hub graph.
STALENESS_BUDGET = 48.0


def auto_syncs(tokens_global, n_words, world):
    """Merges per pass so that (world-1) * tokens per row per interval <= STALENESS_BUDGET."""
    if world <= 1:
        return 1
    return max(1, int(np.ceil(tokens_global * (world - 1) / (STALENESS_BUDGET * max(n_words, 1)))))


def chunk_plan(n_local, n_chunks, exact=False):
    """[begin, end) of every merge interval of a pass over n_local sentences.  exact=True keeps exactly
    n_chunks intervals (some may be empty): every rank must run the same number of collectives even when the
    shards differ in size."""
    n_chunks = max(1, int(n_chunks))
    if not exact:
        n_chunks = min(n_chunks, max(n_local, 1))
    return [shard_bounds(n_local, n_chunks, c) for c in range(n_chunks)]


def hot_every_for(n_local, n_chunks, hot_every="auto", world=2):
    """Hot-tier merges per full interval.  "auto": HOT_EVERY with two replicas — where it brings the hub graph
    inside the AUC band (0.8675 / 0.8672 against 0.8656 / 0.8648 without, comparator 0.8672) — and off beyond:
    with eight replicas it helped at one cadence (+0.003) and hurt at another (-0.004); always
    reduced so that a launch still covers MIN_WALKS_PER_LAUNCH walks."""
    if hot_every != "auto":
        return max(1, int(hot_every))
    if world > 2:
        return 1
    return max(1, min(HOT_EVERY, n_local // (max(n_chunks, 1) * MIN_WALKS_PER_LAUNCH)))


def train(model, walks, lens, epochs=1, comm=None, n_walks_global=None, shard_offset=0, syncs_per_epoch="auto",
          merge="hot", hot_every="auto"):
    """Train `epochs` passes over this rank's walks.  With a communicator the replicas are
    merged `syncs_per_epoch` times per pass ("auto": auto_syncs), the last one at its end; the hot tier
    (TierPlan) `hot_every` times per full interval."""
    n_local = int(walks.shape[0])
    if n_walks_global is None:
        n_walks_global = n_local
    total = epochs * n_walks_global
    world = comm.world if comm is not None else 1
    bases = None
    n_chunks = 1
    if world > 1:
        bases = [model.syn0.clone(), model.syn1neg.clone()] if merge != "avg" else [None, None]
        n_chunks = (auto_syncs(n_walks_global * int(walks.shape[1]), model.n_words, world)
                    if syncs_per_epoch == "auto" else int(syncs_per_epoch))
    # everything that decides how many collectives run must be the same on every rank: derived from the global
    # walk count, never from this rank's shard size
    per_rank = max(1, n_walks_global // world)
    n_chunks = max(1, min(n_chunks, per_rank))
    weights, tier, every = None, None, 1
    if world > 1 and merge == "hot":
        every = hot_every_for(per_rank, n_chunks, hot_every, world)
        if n_chunks * every > per_rank:
            every = 1
        tier = TierPlan(model.counts, n_walks_global * int(walks.shape[1]) / n_chunks, world, model.window,
                        model.negative, model.device, every=every)
        weights, every = tier.w_full, tier.every
    plan = chunk_plan(n_local, n_chunks * every, exact=world > 1)
    for ep in range(epochs):
        for i, (b, e) in enumerate(plan):
            if e > b:
                # all replicas advance together: `b` local sentences = b * world global ones
                model.train_pass(walks[b:e], None if lens is None else lens[b:e],
                                 sentences_base=ep * n_walks_global + b * world, sentences_step=world,
                                 sentences_total=total, walk_id_base=ep * n_walks_global + shard_offset + b)
            if world > 1:
                if (i + 1) % every == 0 or i + 1 == len(plan):
                    merge_replicas([model.syn0, model.syn1neg], bases, comm, merge, weights)
                else:
                    merge_hot_rows([model.syn0, model.syn1neg], bases, comm, tier)
    return model


class _SimulatedComm:
    """all_reduce_sum over replicas that live in ONE process (validation only: the replicas'
    merges are executed one after another on snapshots taken before any of them is changed)."""

    def __init__(self, world, snapshots):
        self.world, self._snap, self._i = world, snapshots, 0

    def all_reduce_sum(self, t):
        total = self._snap[self._i][0].clone()
        for x in self._snap[self._i][1:]:
            total += x
        t.copy_(total)
        self._i += 1


def train_simulated_replicas(models, shards, n_walks_global, syncs_per_epoch="auto", merge="hot", epochs=1,
                             hot_every="auto"):
    """Validation helper: `models` are G replicas on one device, `shards[r] = (walks, lens,
    shard_offset)` what rank r would hold.  Runs the same schedule and the same merge_replicas
    arithmetic as `train`, interval by interval, so the multi-GPU scheme can be scored for AUC
    on a one-GPU box."""
    G = len(models)
    L = int(shards[0][0].shape[1])
    n_chunks = (auto_syncs(n_walks_global * L, models[0].n_words, G) if syncs_per_epoch == "auto"
                else int(syncs_per_epoch))
    bases = [[m.syn0.clone(), m.syn1neg.clone()] if merge != "avg" else [None, None] for m in models]
    per_rank = max(1, n_walks_global // G)
    n_chunks = max(1, min(n_chunks, per_rank))
    weights, tier, every = None, None, 1
    if merge == "hot":
        every = hot_every_for(per_rank, n_chunks, hot_every, G)
        if n_chunks * every > per_rank:
            every = 1
        tier = TierPlan(models[0].counts, n_walks_global * L / n_chunks, G, models[0].window, models[0].negative,
                        models[0].device, every=every)
        weights, every = tier.w_full, tier.every
    plans = [chunk_plan(int(w.shape[0]), n_chunks * every, exact=True) for w, _, _ in shards]
    total = epochs * n_walks_global
    for ep in range(epochs):
        for c in range(len(plans[0])):
            for r, m in enumerate(models):
                w, l, off = shards[r]
                b, e = plans[r][c] if c < len(plans[r]) else (0, 0)
                if e > b:
                    m.train_pass(w[b:e], None if l is None else l[b:e], sentences_base=ep * n_walks_global + b * G,
                                 sentences_step=G, sentences_total=total,
                                 walk_id_base=ep * n_walks_global + off + b)
            if (c + 1) % every == 0 or c + 1 == len(plans[0]):
                snaps = [[m.syn0.clone() for m in models], [m.syn1neg.clone() for m in models]]
                for r, m in enumerate(models):
                    merge_replicas([m.syn0, m.syn1neg], bases[r], _SimulatedComm(G, snaps), merge, weights)
            else:
                names = [n for n, rows in zip(("syn0", "syn1neg"), tier.rows) if rows.numel()]
                snaps = [[getattr(m, n).index_select(0, rows) for m in models]
                         for n, rows in zip(("syn0", "syn1neg"), tier.rows) if rows.numel()]
                assert len(names) == len(snaps)
                for r, m in enumerate(models):
                    merge_hot_rows([m.syn0, m.syn1neg], bases[r], _SimulatedComm(G, snaps), tier)
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
