"""CPU tests of the SGNS host logic: vocabulary statistics vs the oracle's word-by-word
restatement, shard arithmetic, replica merging over gloo (world_size 2), and the device
link-prediction evaluator (run on CPU tensors) vs scikit-learn."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import ROOT


def test_vocab_tables_match_oracle_restatement():
    from n2v_hip import sgns
    from oracle import sgns_oracle
    rs = np.random.RandomState(0)
    for trial in range(5):
        counts = (rs.pareto(1.1, 2000) * 20).astype(np.int64)
        counts[rs.randint(0, 2000, 100)] = 0
        counts[0] = 10**6
        for sample in (1e-3, 1e-5, 0):
            a_si, a_cum = sgns.vocab_tables(counts, sample)
            b_si, b_cum = sgns_oracle.vocab_tables(counts, sample)
            assert (a_si is None) == (b_si is None)
            if a_si is not None:
                assert np.array_equal(a_si, b_si)
            assert np.array_equal(a_cum, b_cum)


def test_shard_bounds_cover_exactly():
    from n2v_hip.sgns import shard_bounds
    for n in (0, 1, 7, 1000, 1000003):
        for world in (1, 2, 3, 8):
            got = [shard_bounds(n, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            assert all(e >= b for b, e in got)


def test_auto_syncs_and_chunk_plan():
    from n2v_hip.sgns import auto_syncs, chunk_plan, merge_constants
    assert auto_syncs(10**9, 10**6, 1) == 1
    assert merge_constants(2) == (512.0, 48.0) and merge_constants(4) == (160.0, 24.0) and merge_constants(8) == (208.0, 48.0)
    assert merge_constants(3) == merge_constants(4) and merge_constants(6) == merge_constants(8) and merge_constants(64) == merge_constants(8)
    for world in (2, 4, 8):
        budget = merge_constants(world)[1]
        k = auto_syncs(30000 * 80, 3000, world)
        assert (world - 1) * (30000 * 80 / k) / 3000 <= budget
        assert (world - 1) * (30000 * 80 / max(k - 1, 1)) / 3000 > budget or k == 1
    plan = chunk_plan(1001, 16)
    assert plan[0][0] == 0 and plan[-1][1] == 1001 and all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
    assert len(chunk_plan(5, 100)) == 5 and chunk_plan(0, 4) == [(0, 0)]
    # exact: the same number of intervals on every rank, empty ones included, covering the shard exactly once
    p5 = chunk_plan(5, 8, exact=True)
    assert len(p5) == 8 and sum(e - b for b, e in p5) == 5 and p5[-1][1] == 5 and len(chunk_plan(0, 3, exact=True)) == 3
    # more intervals than sentences: spread over the whole pass, not packed into the first intervals
    p = chunk_plan(100, 1000, exact=True)
    filled = [i for i, (b, e) in enumerate(p) if e > b]
    assert len(filled) == 100 and max(e - b for b, e in p) == 1 and filled[-1] >= 990 and filled[50] in range(495, 515)


def test_merge_weights_limits():
    from n2v_hip.sgns import merge_weights
    counts = np.array([10**7, 10**3, 1, 0])
    w0, w1 = merge_weights(counts, interval_tokens_global=10**6, world=8, window=10, negative=5, device="cpu")
    assert abs(float(w0[0]) - 1 / 8) < 0.01 and abs(float(w1[0]) - 1 / 8) < 0.01      # hub: mean
    assert float(w0[2]) == 1.0 and float(w0[3]) == 1.0                               # cold rows: sum
    assert (w1 <= w0 + 1e-7).all()                                                   # targets are hotter
    w0, w1 = merge_weights(counts, 10**6, world=1, window=10, negative=5, device="cpu")
    assert (w0 == 1).all() and (w1 == 1).all()
    (s0, s1) = merge_weights(counts, 10**6, 8, 10, 5, "cpu", budget=float("inf"))      # 'delta': pure sum
    (m0, m1) = merge_weights(counts, 10**6, 8, 10, 5, "cpu", budget=0.0)               # 'avg': mean
    assert (s0 == 1).all() and (s1 == 1).all() and torch.allclose(m0, torch.full_like(m0, 1 / 8))


def test_merge_plan_tiers():
    """Rows the other replicas update more than HOT_THETA times per interval form the synchronous tier; the rest
    is merged one interval late.  cold_delay=False puts every row in the synchronous tier."""
    from n2v_hip import sgns
    counts = np.full(1000, 100, dtype=np.int64)
    counts[:5] = 200000                                   # five hubs
    T = counts.sum() / 200.0                              # tokens per interval
    plan = sgns.MergePlan(counts, T, 8, 10, 5, torch.device("cpu"), budget=256.0, theta=256.0, cold_delay=True)
    assert plan.hot_rows[0].tolist() == [0, 1, 2, 3, 4] and set(range(5)) <= set(plan.hot_rows[1].tolist())
    assert plan.hot_pos[0][:5].tolist() == [0, 1, 2, 3, 4] and (plan.hot_pos[0][5:] == -1).all()
    assert plan.n_hot[0] == 5 and plan.n_cold[0] == 995
    (w0, w1), (u0, u1) = sgns.merge_weights(counts, T, 8, 10, 5, torch.device("cpu"), budget=256.0, with_u=True)
    assert torch.equal(plan.w[0], w0) and bool((u0[:5] > 256).all()) and bool((u0[5:] <= 256).all())
    assert (plan.w[0][5:] == 1).all() and (plan.w[0][:5] < 0.2).all()
    sync = sgns.MergePlan(counts, T, 8, 10, 5, torch.device("cpu"))        # default: no delayed tier
    assert sync.n_hot == [1000, 1000] and sync.n_cold == [0, 0]
    avg = sgns.MergePlan(counts, T, 8, 10, 5, torch.device("cpu"), mode="avg")
    assert torch.allclose(avg.w[0], torch.full((1000,), 1 / 8))
    with pytest.raises(ValueError):
        sgns.MergePlan(counts, T, 8, 10, 5, torch.device("cpu"), mode="sparse")


def _fake_plan(w, hot_rows, n):
    """A MergePlan with hand-made tiers: one table, `hot_rows` synchronous."""
    from n2v_hip import sgns
    plan = sgns.MergePlan.__new__(sgns.MergePlan)
    rows = torch.tensor(hot_rows, dtype=torch.int64)
    pos = torch.full((n,), -1, dtype=torch.int32)
    pos[rows] = torch.arange(len(hot_rows), dtype=torch.int32)
    plan.w, plan.hot_rows, plan.hot_pos = [w], [rows], [pos]
    plan.n_hot, plan.n_cold, plan.world, plan.cold_delay = [len(hot_rows)], [n - len(hot_rows)], 2, True
    return plan


def test_merger_protocol_is_linear_in_the_changes():
    """When the replicas' changes do not depend on the tables (here: fixed increments), every schedule must end
    with base0 + sum_k w * sum_r d_rk on every replica: two tiers, the cold rows one interval late, overlap on
    or off, float32 or bfloat16 wire."""
    from merge_reference import TorchMergeOps
    from n2v_hip import sgns
    G, n, stride, K = 3, 7, 4, 5
    g = torch.Generator().manual_seed(1)
    base0 = torch.randn(n, stride, generator=g)
    w = torch.tensor([1.0, 1.0, 0.5, 0.25, 1.0, 0.75, 1.0])
    incr = torch.randn(K, G, n, stride, generator=g) * 0.1
    for wire in (None, torch.bfloat16):
        for hot_rows in ([2, 3, 5], [], list(range(n))):
            plan = _fake_plan(w, hot_rows, n)
            group = sgns._SimGroup(G, wire)
            tabs = [base0.clone() for _ in range(G)]
            mergers = [sgns.ReplicaMerger([t], plan, group.comm(), ops=TorchMergeOps()) for t in tabs]
            for k in range(K):
                for r in range(G):
                    tabs[r] += incr[k, r]
                for mg in mergers:
                    mg.snapshot()
                if mergers[0].hot_wire is not None:
                    sgns._SimGroup.reduce([mg.hot_wire for mg in mergers])
                for mg in mergers:
                    mg.finish()
            for mg in mergers:
                mg.flush()
            want = base0 + w[:, None] * incr.sum(dim=(0, 1))
            tol = 1e-5 if wire is None else 2e-2
            for t in tabs:
                assert torch.allclose(t, want, atol=tol), (wire, hot_rows, (t - want).abs().max())
                assert torch.equal(t, tabs[0])
            assert all(torch.equal(mg.base[0], tabs[0]) and torch.equal(mg.xs[0], tabs[0]) for mg in mergers)


def test_sum_tier_plan_and_schedule():
    """merge="tsum": tier j rows are merged ratio^j times per base interval so that no row collects more than theta
    foreign updates between two of its merges."""
    from n2v_hip import sgns
    counts = np.full(1000, 100, dtype=np.int64)
    counts[:3] = 3000          # 30 x the average
    counts[3] = 60000          # 600 x: beyond the last tier
    T = counts.sum() * 0.2     # tokens per base interval
    plan = sgns.SumTierPlan(counts, T, 8, 10, 5, torch.device("cpu"), theta=500.0, n_tiers=4, ratio=4)   # explicit theta
    (u0, u1) = [(7 / 8) * x for x in sgns.expected_updates(counts, T, 10, 5, torch.device("cpu"))]
    assert plan.sub == 64
    for ti, u in enumerate((u0, u1)):
        t = plan.tier[ti]
        assert bool((t[u <= 500] == 0).all())
        for j in (1, 2):
            sel = (u > 500 * 4 ** (j - 1)) & (u <= 500 * 4 ** j)
            assert bool((t[sel] == j).all())
        assert bool((t[u > 500 * 16] == 3).all())
        # between two merges of a row (except beyond the last tier): at most theta foreign updates
        per_merge = u / (4.0 ** t.double())
        assert bool((per_merge[u <= 500 * 64] <= 500 + 1e-9).all())
        assert plan.rows_ge[ti][0].numel() == 1000 and plan.rows_ge[ti][3].tolist() == torch.nonzero(t >= 3).flatten().tolist()
    flat = sgns.SumTierPlan(np.full(1000, 100, dtype=np.int64), T, 8, 10, 5, torch.device("cpu"), theta=1e9)
    assert flat.n_tiers == 1 and flat.sub == 1 and flat.level_due(0) == 0     # no hubs: one tier, no sub-intervals
    due = [plan.level_due(c) for c in range(128)]
    assert due[63] == 0 and due[127] == 0 and due[15] == 1 and due[31] == 1 and due[3] == 2 and due[0] == 3
    assert sum(1 for d in due[:64] if d is not None and d <= 1) == 4 and all(d is not None for d in due)   # ratio^(tiers-1) levels: every sub-interval merges the last tier


def test_tiered_sum_merger_applies_every_change_once():
    """Pure sums at per-row cadences: whatever the tiers, after the pass every replica holds
    base0 + (sum of all changes of all replicas), and the replicas agree after every level-0 merge."""
    from merge_reference import TorchMergeOps
    from n2v_hip import sgns
    G, n, stride = 3, 9, 4
    g = torch.Generator().manual_seed(2)
    base0 = torch.randn(n, stride, generator=g)
    plan = sgns.SumTierPlan.__new__(sgns.SumTierPlan)
    plan.n_tiers, plan.ratio, plan.sub, plan.world = 3, 2, 4, G
    tier = torch.tensor([0, 0, 1, 2, 0, 1, 0, 2, 0])
    plan.tier = [tier]
    plan.rows_ge = [[torch.arange(n)] + [torch.nonzero(tier >= j).flatten() for j in (1, 2)]]
    n_sub = 12                                   # 3 base intervals x 4 sub-intervals
    incr = torch.randn(n_sub, G, n, stride, generator=g) * 0.1
    for wire in (None, torch.bfloat16):
        group = sgns._SimGroup(G, wire)
        tabs = [base0.clone() for _ in range(G)]
        mergers = [sgns.TieredSumMerger([t], plan, group.comm(), ops=TorchMergeOps()) for t in tabs]
        for c in range(n_sub):
            for r in range(G):
                tabs[r] += incr[c, r]
            level = plan.level_due(c)
            assert level is not None
            packed = [mg.pack(level) for mg in mergers]
            sgns._SimGroup.reduce([flat for flat, _ in packed])
            for mg, (_, v) in zip(mergers, packed):
                mg.apply(level, v)
            hot = plan.rows_ge[0][level]
            for t in tabs[1:]:
                assert torch.equal(t[hot], tabs[0][hot])          # merged rows agree on every replica
            if level == 0:
                want = base0 + incr[: c + 1].sum(dim=(0, 1))
                assert torch.allclose(tabs[0], want, atol=1e-5 if wire is None else 3e-2)
        assert mergers[0].n_merges == [3, 3, 6]


def test_cold_rows_keep_the_own_change_until_it_is_merged():
    """After an interval a replica's cold rows hold base + its OWN change (the other replicas' changes arrive one
    interval later); its hot rows are merged at once."""
    from merge_reference import TorchMergeOps
    from n2v_hip import sgns
    n, stride = 4, 2
    base0 = torch.zeros(n, stride)
    plan = _fake_plan(torch.ones(n), [0], n)
    group = sgns._SimGroup(2, None)
    tabs = [base0.clone(), base0.clone()]
    mergers = [sgns.ReplicaMerger([t], plan, group.comm(), ops=TorchMergeOps()) for t in tabs]
    tabs[0] += 1.0
    tabs[1] += 10.0
    for mg in mergers:
        mg.snapshot()
    sgns._SimGroup.reduce([mg.hot_wire for mg in mergers])
    for mg in mergers:
        mg.finish()
    assert tabs[0][0].tolist() == [11.0, 11.0] and tabs[1][0].tolist() == [11.0, 11.0]     # hot row: merged
    assert tabs[0][1].tolist() == [1.0, 1.0] and tabs[1][1].tolist() == [10.0, 10.0]       # cold rows: own change only
    for mg in mergers:          # an interval without training: the late sums arrive
        mg.snapshot()
    assert tabs[0][1].tolist() == [11.0, 11.0] and tabs[1][1].tolist() == [11.0, 11.0]


def test_linkpred_metrics_match_sklearn():
    from sklearn.metrics import average_precision_score, roc_auc_score
    from n2v_hip import linkpred
    rs = np.random.RandomState(0)
    for trial in range(5):
        pos = np.round(rs.normal(0.3, 1, 400), 1 if trial % 2 else 6)   # with and without ties
        neg = np.round(rs.normal(0.0, 1, 700), 1 if trial % 2 else 6)
        y = np.r_[np.ones(400), np.zeros(700)]
        s = np.r_[pos, neg]
        a = linkpred.roc_auc(torch.from_numpy(pos), torch.from_numpy(neg))
        b = linkpred.average_precision(torch.from_numpy(pos), torch.from_numpy(neg))
        assert abs(a - roc_auc_score(y, s)) < 1e-12
        assert abs(b - average_precision_score(y, s)) < 1e-12
    v = torch.from_numpy(rs.normal(size=(50, 16)).astype(np.float32))
    pairs = torch.from_numpy(rs.randint(0, 50, size=(30, 2)))
    got = linkpred.cosine_scores(v, pairs).numpy()
    want = [float(np.dot(v[a] / np.linalg.norm(v[a]), v[b] / np.linalg.norm(v[b]))) for a, b in pairs.tolist()]
    np.testing.assert_allclose(got, want, rtol=1e-5)


_GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "node2vec-by-ecc_amd"))
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import torch, torch.distributed as dist
from n2v_hip import sgns
from merge_reference import TorchMergeOps
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
comm = sgns._ProcessGroupComm()
assert comm.world == 2
n, stride, K = 6, 4, 4
g = torch.Generator().manual_seed(5)
base0 = torch.randn(n, stride, generator=g)
w = torch.tensor([1.0, 1.0, 0.5, 0.5, 1.0, 0.75])
incr = torch.randn(K, 2, n, stride, generator=g)
plan = sgns.MergePlan.__new__(sgns.MergePlan)
rows = torch.tensor([2, 3])
pos = torch.full((n,), -1, dtype=torch.int32); pos[rows] = torch.arange(2, dtype=torch.int32)
plan.w, plan.hot_rows, plan.hot_pos, plan.n_hot, plan.n_cold, plan.world, plan.cold_delay = [w], [rows], [pos], [2], [4], 2, True
results = []
for overlap in (True, False):
    t = base0.clone()
    mg = sgns.ReplicaMerger([t], plan, comm, overlap=overlap, ops=TorchMergeOps(), pipe=3, pipe_bytes=32)
    for k in range(K):
        t += incr[k, rank]                     # this rank's "training" of interval k
        mg.end_interval(last=(k + 1 == K))
    want = base0 + w[:, None] * incr.sum(dim=(0, 1))
    assert torch.allclose(t, want, atol=1e-5), (overlap, t, want)
    other = t.clone(); dist.broadcast(other, src=0)
    assert torch.equal(other, t)               # identical tables on both ranks
    assert mg.n_merges == K
    results.append(t.clone())
assert torch.equal(results[0], results[1])     # the same bits with the cold all-reduce overlapped or waited for
# tiered pure sums over the same process group
tp = sgns.SumTierPlan.__new__(sgns.SumTierPlan)
tp.n_tiers, tp.ratio, tp.sub, tp.world = 2, 2, 2, 2
tier = torch.tensor([0, 1, 0, 1, 0, 0])
tp.tier = [tier]
tp.rows_ge = [[torch.arange(n), torch.nonzero(tier >= 1).flatten()]]
t = base0.clone()
mg = sgns.TieredSumMerger([t], tp, comm, ops=TorchMergeOps())
for k in range(K):
    t += incr[k, rank]
    mg.merge(tp.level_due(k))
assert torch.allclose(t, base0 + incr.sum(dim=(0, 1)), atol=1e-5)
other = t.clone(); dist.broadcast(other, src=0)
assert torch.equal(other, t) and mg.n_merges == [K // 2, K // 2]
b, e = sgns.shard_bounds(101, 2, rank)
tot = torch.tensor([e - b]); dist.all_reduce(tot); assert int(tot) == 101
# the whole driver (sgns.train, merge="tsum") with UNEVEN shards and a stand-in model whose "training" of a walk
# adds 1 to the row of each of its tokens: same number of collectives on both ranks (no deadlock), every walk applied
# exactly once, identical tables at the end
import numpy as np
class StandIn:
    n_words, window, negative, device = 40, 10, 5, torch.device("cpu")
    def __init__(self):
        self.counts = np.r_[np.full(4, 4000), np.full(36, 50)].astype(np.int64)      # four hub rows -> more than one tier
        self.syn0 = torch.zeros(40, 4); self.syn1neg = torch.zeros(40, 4)
        self.trained = []
    def span_trainer(self, walks, lens, sentences_total, sentences_step, splits="auto"):
        def launch(b, e, sentences_base, walk_id_base):
            for w in range(b, e):
                self.trained.append(walk_id_base + (w - b))
                for tok in walks[w].tolist():
                    self.syn0[tok] += 1.0; self.syn1neg[tok] -= 0.5
        return launch
    def train_pass(self, walks, lens, sentences_base, sentences_total, walk_id_base, sentences_step=1, max_blocks=0, splits=1):
        self.span_trainer(walks, lens, sentences_total, sentences_step)(0, int(walks.shape[0]), sentences_base, walk_id_base)
g = torch.Generator().manual_seed(9)
n_global = 61
all_walks = torch.randint(0, 40, (n_global, 8), generator=g, dtype=torch.int32)
b, e = sgns.shard_bounds(n_global, 2, rank)                                           # 31 and 30 walks
m = StandIn()
mg = sgns.train(m, all_walks[b:e].contiguous(), None, epochs=1, comm=comm, n_walks_global=n_global, shard_offset=b,
                merge="tsum", ops=TorchMergeOps())
assert mg.plan.n_tiers >= 2 and sum(mg.n_merges) >= mg.plan.sub, (mg.plan.n_tiers, mg.n_merges)
assert m.trained == list(range(b, e))                                                # every local walk once, in order
want = torch.zeros(40, 4)
for tok in all_walks.reshape(-1).tolist():
    want[tok] += 1.0
assert torch.equal(m.syn0, want) and torch.equal(m.syn1neg, -0.5 * want), (m.syn0[:5], want[:5])
other = m.syn0.clone(); dist.broadcast(other, src=0); assert torch.equal(other, m.syn0)
# the weighted merges through the same driver: linear in the changes, weight w[row] on the summed change
m2 = StandIn()
mg2 = sgns.train(m2, all_walks[b:e].contiguous(), None, epochs=1, comm=comm, n_walks_global=n_global, shard_offset=b,
                 merge="hot", syncs_per_epoch=5, ops=TorchMergeOps())
assert m2.trained == list(range(b, e)) and mg2.n_merges == 5
assert torch.allclose(m2.syn0, mg2.plan.w[0][:, None] * want, atol=1e-4), (m2.syn0[:5], want[:5], mg2.plan.w[0][:5])
assert torch.allclose(m2.syn1neg, -0.5 * mg2.plan.w[1][:, None] * want, atol=1e-4)
other = m2.syn0.clone(); dist.broadcast(other, src=0); assert torch.equal(other, m2.syn0)
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_replica_merge_over_gloo_world2(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT, "port": port})
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_build_neg_samples_and_isolated_nodes():
    from n2v_hip import csr, linkpred
    rs = np.random.RandomState(0)
    labels = np.arange(0, 400, 2)                                     # 200 nodes, even labels
    e = labels[rs.randint(0, 200, size=(3000, 2))]
    e = e[e[:, 0] != e[:, 1]]
    neg = linkpred.build_neg_samples(labels, e, seed=1)
    true = set((min(a, b), max(a, b)) for a, b in e.tolist())
    assert len(neg) == len(true)
    got = set(map(tuple, neg.tolist()))
    assert len(got) == len(neg) and not (got & true) and all(a < b for a, b in got)
    full = csr.from_edges([0, 2, 4, 6], [2, 4, 0, 8])
    train = csr.from_edges([0, 2], [2, 4])                             # 6 and 8 lost their edges
    t2 = linkpred._with_isolated_nodes(train, full)
    assert t2.n_nodes == 5 and t2.degrees.tolist() == [1, 2, 1, 0, 0]
    assert t2.labels[t2.col].tolist() == [2, 0, 4, 2] and np.array_equal(t2.start_order, full.start_order)


def test_auto_syncs_c3_shapes():
    from n2v_hip import sgns
    # C3: 10M walks of 80 over 1M rows; 2 ranks -> 34 merges per pass, 8 ranks -> 934 at STALENESS_BUDGET 48
    assert sgns.auto_syncs(10_000_000 * 80, 1_000_000, 2) == 17
    assert sgns.auto_syncs(10_000_000 * 80, 1_000_000, 8) == 117
    assert sgns.auto_syncs(80_000_000 * 80, 1_000_000, 8) == 934
    assert sgns.auto_syncs(10_000_000 * 80, 1_000_000, 4) == 100


def test_merger_without_gpu_has_no_fallback():
    from n2v_hip import sgns
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    plan = _fake_plan(torch.ones(3), [0], 3)
    mg = sgns.ReplicaMerger([torch.zeros(3, 2)], plan, sgns._SimGroup(2).comm())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        mg.snapshot()


def test_out_of_band_modes_are_fenced():
    """Modes that were MEASURED outside BASELINE.json's +-0.002 AUC band cannot be reached by accident:
    the `auto` row-sharing rule, explicit lossy row sharing on short corpora / small tables, several wavefronts per
    sentence with lossy rows, merge="hot" on large vocabularies and shared negatives all need allow_out_of_band=True.
    Pure host logic: runs without a GPU."""
    import pytest
    from n2v_hip import sgns
    n = sgns.AUTO_AGENT_MIN_WORDS
    for rows, tpr, want in ((n, 800, "agent"), (n, 200, "atomic"), (n - 1, 800, "atomic"), (3000, 800, "atomic"),
                            (4 * n, 600, "agent"), (4 * n, 599, "atomic")):
        assert sgns.resolve_update_mode("auto", rows, rows * tpr) == want, (rows, tpr)
    for mode in ("agent", "plain"):
        assert sgns.resolve_update_mode(mode, n, 800 * n) == mode                  # long corpus, large table: in the band
        for rows, tpr in ((n, 200), (n, 5), (3000, 800)):
            with pytest.raises(sgns.OutOfBandError):
                sgns.resolve_update_mode(mode, rows, rows * tpr)
            assert sgns.resolve_update_mode(mode, rows, rows * tpr, allow_out_of_band=True) == mode
    assert sgns.resolve_update_mode("atomic", 10, 10) == "atomic"
    with pytest.raises(ValueError):
        sgns.resolve_update_mode("fast", n, 800 * n)
    # walk_splits > 1: the lossless mode, silently under `auto`, by request otherwise
    A, G, P = sgns.UPDATE_MODES["atomic"], sgns.UPDATE_MODES["agent"], sgns.UPDATE_MODES["plain"]
    assert sgns.launch_update_mode(G, True, 8) == A and sgns.launch_update_mode(P, True, 2) == A
    assert sgns.launch_update_mode(G, True, 1) == G and sgns.launch_update_mode(A, False, 80) == A
    with pytest.raises(sgns.OutOfBandError):
        sgns.launch_update_mode(G, False, 8)
    assert sgns.launch_update_mode(G, False, 8, allow_out_of_band=True) == G | 8   # N2V_SGNS_UNCHECKED
    # several wavefronts per sentence only on large tables (the short launches' synchronous starts bias small graphs)
    assert sgns.auto_splits(20000, 13, 80) == 1 and sgns.auto_splits(sgns.AUTO_SPLITS_MIN_WORDS, 13, 80) == 80
    assert sgns.auto_splits(10**6, 83, 80) == 80 and sgns.auto_splits(10**6, 1000, 80) == 9 and sgns.auto_splits(10**6, 10**5, 80) == 1
    # merge="hot" above the size it was shown to hold at
    sgns.check_merge_in_band("tsum", 10**7)
    sgns.check_merge_in_band("hot", 20000)
    with pytest.raises(sgns.OutOfBandError):
        sgns.check_merge_in_band("hot", 131072)
    sgns.check_merge_in_band("hot", 131072, allow_out_of_band=True)
    with pytest.raises(ValueError):
        sgns.check_merge_in_band("avg", 10)
    # shared negatives: refused before anything touches a device
    with pytest.raises(sgns.OutOfBandError):
        sgns.SgnsModel(10, share_negatives=True)
    with pytest.raises(ValueError):
        sgns.SgnsModel(10, update_mode="fast")
    import main as n2v_main
    assert n2v_main.parse_args(["--input", "x"]).allow_out_of_band is False
    assert n2v_main.parse_args(["--input", "x", "--allow-out-of-band", "--merge", "hot"]).allow_out_of_band is True


def test_default_grid_is_whole_workgroups_per_cu():
    """n2v_sgns_train's default grid (include/n2v_hip.h): at most one wavefront per 64 vocabulary rows, never more than 3 072 workgroups, and a whole number of workgroups per CU once there is
    more than one — 1 561 workgroups on a 399 846-row table cost 0.004 of AUC."""
    from n2v_hip import _lib
    lib = _lib.load()
    cus = 256                                            # without a GPU the library assumes MI355X's 256 CUs
    for n in (10, 3000, 19998, 65535, 131019, 131072, 200000, 399846, 600000, 10**6, 10**8):
        for mode, rows_per_wave in ((_lib.N2V_SGNS_AGENT if hasattr(_lib, "N2V_SGNS_AGENT") else 1, 64), (2, 64)):
            b = lib.n2v_sgns_default_blocks(n, mode)
            assert 16 <= b <= 3072
            assert b <= max(16, n // 256) or (mode != 2 and b == 4 * cus)   # store-based rows: at least 4 workgroups per CU
            assert b <= cus or b % cus == 0, (n, mode, b)
    assert lib.n2v_sgns_default_blocks(399846, 1) == 1536 and lib.n2v_sgns_default_blocks(399846, 2) == 1536
    assert lib.n2v_sgns_default_blocks(10**6, 1) == 3072 and lib.n2v_sgns_default_blocks(10**6, 2) == 3072
    assert lib.n2v_sgns_default_blocks(131019, 1) == 1024 and lib.n2v_sgns_default_blocks(131019, 2) == 256
