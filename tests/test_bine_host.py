"""CPU tests of the BiNE restatements (oracle/bine_oracle.py) and the host-side graph builder.

The reference's BiNE files cannot run here (see the oracle's header), so the chain is closed statistically:
the Philox restatement (P) — which the HIP kernels must equal bit for bit (tests/test_gpu_bine.py) — is
checked against the literal restatement (L) of the reference text: next-vertex distribution uniform over the
DISTINCT two-hop vertices, geometric walk lengths, HITS direction."""
import math
import os
import random
import sys

import numpy as np
import pytest
from scipy import stats

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))

from oracle import bine_oracle as bo  # noqa: E402


def small_graph(seed=3, n_u=30, n_v=12, per_user=5):
    from n2v_hip import bine
    rs = np.random.RandomState(seed)
    users = np.repeat(np.arange(n_u), per_user)
    items = rs.randint(0, n_v, size=users.shape[0])       # dense overlaps: multiplicities well above 1
    ratings = rs.randint(1, 6, size=users.shape[0]).astype(float)
    return bine.BipartiteGraph(["u%d" % u for u in users], ["i%d" % i for i in items], ratings)


def test_bipartite_graph_builder():
    from n2v_hip import bine
    g = bine.BipartiteGraph(["u2", "u10", "u2", "u2", "u3"], ["i1", "i1", "i5", "i1", "i5"], [1, 2, 3, 4, 5])
    assert g.user_labels.tolist() == ["u10", "u2", "u3"]        # string sort, as node_u.sort()
    assert g.item_labels.tolist() == ["i1", "i5"]
    assert (g.n_u, g.n_v, g.n_ratings) == (3, 2, 5)
    assert g.edge_u.tolist() == [1, 0, 1, 1, 2] and g.edge_v.tolist() == [3, 3, 4, 3, 4]
    assert g.edge_w.tolist() == [4.0, 2.0, 3.0, 4.0, 5.0]       # the repeated (u2, i1) takes its last rating
    assert g.first.tolist() == [3, 1, 2, 0, 1]
    # symmetric CSR, rows ascending
    assert g.row_ptr.tolist() == [0, 1, 3, 4, 6, 8]
    assert g.col.tolist() == [3, 3, 4, 4, 0, 1, 1, 2]
    assert g.w.tolist() == [2.0, 4.0, 3.0, 5.0, 2.0, 4.0, 3.0, 5.0]


def test_hits_restatement_is_the_top_singular_pair():
    g = small_graph()
    a, iters = bo.hits_nx111(g.row_ptr, g.col, g.w)
    B = np.zeros((g.n_u, g.n_v))
    for e in range(g.n_ratings):
        B[g.edge_u[e], g.edge_v[e] - g.n_u] = g.edge_w[e]
    U, S, Vt = np.linalg.svd(B)
    for seg, sv in ((a[: g.n_u], np.abs(U[:, 0])), (a[g.n_u:], np.abs(Vt[0]))):
        assert np.allclose(seg / seg.max(), sv / sv.max(), rtol=0, atol=1e-6)
    counts, auth = bo.walk_counts(a, 0, g.n_u, 32, 1)
    assert counts.min() >= 1 and counts.max() == 32 and auth.min() == 0.0 and auth.max() == 1.0


def test_first_common_neighbour_thinning_is_uniform_over_distinct_vertices():
    """(P) vs (L): from a fixed vertex the next vertex must be uniform over the distinct two-hop vertices
    (the reference's rand.choice over a de-duplicated matrix row), NOT weighted by the number of paths."""
    g = small_graph()
    cum2 = bo.two_hop_prefix(g.row_ptr, g.col)
    for start in (0, 7, g.n_u + 2):
        distinct = [x for x in bo.projection_rows(g.row_ptr, g.col, start) if x != start]
        # multiplicities must be non-trivial for this to test anything
        mult = [len(set(g.col[g.row_ptr[start]:g.row_ptr[start + 1]]) & set(g.col[g.row_ptr[x]:g.row_ptr[x + 1]]))
                for x in distinct]
        assert max(mult) > 1
        n = 6000
        got = {x: 0 for x in distinct}
        for gw in range(n):
            got[bo.device_walk(g.row_ptr, g.col, cum2, start, gw, 2, seed=12345 + start)[1]] += 1
        chi2, p = stats.chisquare(list(got.values()))
        assert p > 1e-3, (start, chi2, p)
        # and the literal restatement agrees with itself on the same test (sanity of the yardstick)
        rand = random.Random(7)
        lit = {x: 0 for x in distinct}
        rows = lambda v: bo.projection_rows(g.row_ptr, g.col, v)  # noqa: E731
        for _ in range(n):
            lit[bo.literal_walk(rows, start, -1.0, rand, max_tokens=2)[1]] += 1   # always continue, one step
        assert stats.chi2_contingency([list(got.values()), list(lit.values())])[1] > 1e-3


def test_walk_lengths_are_geometric():
    g = small_graph()
    cum2 = bo.two_hop_prefix(g.row_ptr, g.col)
    p = 0.15
    lens = np.array([bo.walk_length(g.row_ptr, cum2, 0, gw, p, 256, seed=99) for gw in range(20000)])
    # P(len = k) = p (1-p)^(k-1)
    ks = np.arange(1, 25)
    exp = len(lens) * p * (1 - p) ** (ks - 1)
    obs = np.array([(lens == k).sum() for k in ks])
    chi2 = ((obs - exp) ** 2 / exp).sum()
    assert stats.chi2.sf(chi2, len(ks) - 1) > 1e-3
    rand = random.Random(1)
    lit = np.array([len(bo.literal_walk(lambda v: bo.projection_rows(g.row_ptr, g.col, v), 0, p, rand))
                    for _ in range(4000)])
    assert abs(lit.mean() - lens.mean()) < 0.35


def test_dead_end_start_yields_single_token():
    from n2v_hip import bine
    g = bine.BipartiteGraph(["u0", "u1", "u1"], ["i0", "i1", "i2"], [1, 1, 1])   # u0 alone on i0
    cum2 = bo.two_hop_prefix(g.row_ptr, g.col)
    assert bo.walk_length(g.row_ptr, cum2, 0, 0, 0.0, 256, 1) == 1
    assert bo.walk_length(g.row_ptr, cum2, g.n_u + 1, 0, 0.0, 8, 1) == 8       # i1 - u1 - i2: keeps walking
    assert bo.device_walk(g.row_ptr, g.col, cum2, g.n_u + 1, 0, 4, 5) == [3, 4, 3, 4]


def test_negative_pool_excludes_self_and_similar():
    g = small_graph(n_u=60, n_v=40, per_user=3)
    for v in (0, 5, g.n_u + 1):
        lo, hi = (0, g.n_u) if v < g.n_u else (g.n_u, g.n)
        pool = bo.neg_pool(g.row_ptr, g.col, lo, hi, v, 50, 0.2, seed=4)
        assert all(lo <= c < hi and c != v for c in pool)
        nv = set(g.col[g.row_ptr[v]:g.row_ptr[v + 1]])
        jac = [len(nv & set(g.col[g.row_ptr[c]:g.row_ptr[c + 1]])) / len(nv | set(g.col[g.row_ptr[c]:g.row_ptr[c + 1]]))
               for c in pool]
        assert np.mean(np.array(jac) <= 0.2) > 0.95   # the 16-draw thinning may give up, rarely


def test_training_restatement_runs_and_steps_lambda():
    g = small_graph(n_u=20, n_v=10, per_user=4)
    cum2 = bo.two_hop_prefix(g.row_ptr, g.col)
    a, _ = bo.hits_nx111(g.row_ptr, g.col, g.w)
    tokens, tok_walk, walk_off = [], [], [0]
    for lo, hi, seed in ((0, g.n_u, 1), (g.n_u, g.n, 2)):
        counts, _ = bo.walk_counts(a, lo, hi, 4, 1)
        gw = 0
        for v in range(lo, hi):
            for _ in range(int(counts[v - lo])):
                L = bo.walk_length(g.row_ptr, cum2, v, gw, 0.15, 256, seed)
                wk = bo.device_walk(g.row_ptr, g.col, cum2, v, gw, L, seed)
                tok_walk.extend([len(walk_off) - 1] * len(wk))
                tokens.extend(wk)
                walk_off.append(len(tokens))
                gw += 1
    tokens = np.array(tokens)
    order = np.argsort(tokens, kind="stable")
    occ_ptr = np.concatenate([[0], np.cumsum(np.bincount(tokens, minlength=g.n))])
    pool = [bo.neg_pool(g.row_ptr, g.col, *((0, g.n_u) if v < g.n_u else (g.n_u, g.n)), v, 8, 0.5, 3) for v in range(g.n)]
    emb, ctx = bo.init_rows(g.n, 8, 5)
    assert np.allclose((emb ** 2).sum(1), 1.0) and emb.min() >= 0
    lam, losses = bo.train(g.edge_u, g.edge_v, g.edge_w, emb, ctx, occ_ptr, order, tokens, np.array(tok_walk),
                           np.array(walk_off), pool, 5, 4, 0.01, 0.01, 0.1, 0.01, 6, 11, 12)
    assert len(losses) == 6 and all(math.isfinite(x) for x in losses)
    # the learning rate follows the loss-driven rule of src/bine_train.py:495-500
    want, last = 0.01, 0.0
    for x in losses:
        want = want * 1.05 if last > x else want * 0.95
        last = x
    assert lam == want
    e0, _ = bo.init_rows(g.n, 8, 5)
    assert np.abs(emb - e0).max() > 1e-4


_GLOO_WORKER = r'''
import os, sys, types
sys.path.insert(0, os.path.join(%(root)r, "node2vec-by-ecc_amd"))
import torch, torch.distributed as dist
from n2v_hip import bine, sgns
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
comm = sgns._ProcessGroupComm()
base = torch.arange(24, dtype=torch.float64).reshape(4, 6)
eng = types.SimpleNamespace(emb=base.clone(), ctx=-base.clone(), state=torch.zeros(8, dtype=torch.float64))
merge = bine.ReplicaMerge(eng, comm)
for step in range(2):
    eng.emb[rank] += 1.0 + rank          # each replica changes its own row ...
    eng.emb[3] += 10.0 * (rank + 1)      # ... and both change row 3
    eng.ctx[2] -= 0.5
    eng.state[1] = -(3.0 + rank)         # this replica's part of the loss
    merge(eng)
    want = base.clone(); want[0] += 1.0 * (step + 1); want[1] += 2.0 * (step + 1); want[3] += 30.0 * (step + 1)
    assert torch.equal(eng.emb, want), (eng.emb, want)
    wc = -base.clone(); wc[2] -= 1.0 * (step + 1)
    assert torch.equal(eng.ctx, wc)
    assert eng.state[1].item() == -7.0
    assert torch.equal(merge.base[0], eng.emb) and torch.equal(merge.base[1], eng.ctx)
# the overlapped merge (all-reduce of pass i under pass i+1): same totals as the synchronous one, bit for bit, for
# changes that do not depend on what the other rank did (integer-valued here, so every sum is exact); between the
# passes a rank sees the other's changes ONE PASS LATE, and flush() settles the last one
eng_s = types.SimpleNamespace(emb=base.clone(), ctx=-base.clone(), state=torch.zeros(8, dtype=torch.float64))
eng_o = types.SimpleNamespace(emb=base.clone(), ctx=-base.clone(), state=torch.zeros(8, dtype=torch.float64))
sync, over = bine.ReplicaMerge(eng_s, comm), bine.OverlappedReplicaMerge(eng_o, comm)
for step in range(3):
    for eng in (eng_s, eng_o):
        eng.emb[rank] += 1.0 + rank + step
        eng.emb[3] += 10.0 * (rank + 1)
        eng.ctx[2] -= 4.0 + step
        eng.state[1] = -(3.0 + rank + step)
    sync(eng_s)
    over(eng_o)
    assert eng_o.state[1].item() == eng_s.state[1].item() == -(7.0 + 2 * step)      # the loss is never late
    other = 1 - rank
    late = eng_s.emb.clone()
    late[other] -= 1.0 + other + step                  # the other rank's change of THIS pass has not arrived yet
    late[3] -= 10.0 * (other + 1)
    assert torch.equal(eng_o.emb, late), (step, eng_o.emb, late)
over.flush(eng_o)
assert torch.equal(eng_o.emb, eng_s.emb) and torch.equal(eng_o.ctx, eng_s.ctx)
assert torch.equal(over.xs[0], eng_o.emb) and over.pending is None
chk = eng_o.emb.clone(); dist.broadcast(chk, src=0); assert torch.equal(chk, eng_o.emb)
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_bine_replica_merge_over_gloo_world2(tmp_path):
    import subprocess
    port = 31500 + (os.getpid() % 2000)
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT, "port": port})
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_recommendation_metrics_match_the_reference_formulas():
    """top_N's metric helpers (src/bine_train.py:361-406) on a hand-checked case."""
    import bine_train as bt
    ranked, truth = ["a", "b", "c", "d"], ["b", "d", "x"]
    assert bt.precision_and_racall(ranked, truth) == (0.5, 2 / 3)
    assert bt.AP(ranked, truth) == pytest.approx((1 / 2 + 2 / 4) / 3)
    assert bt.RR(ranked, truth) == 0.5
    idcg = 1 / math.log(2, 2) + 1 / math.log(3, 2) + 1 / math.log(4, 2)
    assert bt.nDCG(ranked, truth) == pytest.approx((1 / math.log(3, 2) + 1 / math.log(5, 2)) / idcg)
    assert bt.ndarray_tostring(np.array([[0.5, 1.0]])) == "0.5 1.0 \n"
    a = bt.default_args(d=64)
    assert (a.ws, a.ns, a.maxT, a.minT, a.p, a.alpha, a.beta, a.gamma, a.lam, a.max_iter, a.d) == \
        (5, 4, 32, 1, 0.15, 0.01, 0.01, 0.1, 0.01, 50, 64)


def test_bine_engine_has_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from n2v_hip import bine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        bine.BineEngine(small_graph())


def test_bine_replica_merge_with_fp32_wire():
    import types
    import torch
    from n2v_hip import bine

    class TwoIdenticalReplicas:
        world, wire_dtype_f64 = 2, torch.float32

        def all_reduce_sum(self, t):
            t.mul_(2)

    base = torch.arange(12, dtype=torch.float64).reshape(3, 4)
    eng = types.SimpleNamespace(emb=base.clone(), ctx=base.clone(), state=torch.zeros(8, dtype=torch.float64))
    m = bine.ReplicaMerge(eng, TwoIdenticalReplicas())
    eng.emb += 1e-3
    eng.ctx[1] -= 0.25
    eng.state[1] = -2.0
    m(eng)
    assert torch.allclose(eng.emb, base + 2e-3, rtol=0, atol=1e-9) and eng.emb.dtype == torch.float64
    want = base.clone(); want[1] -= 0.5
    assert torch.equal(eng.ctx, want) and eng.state[1].item() == -4.0
