"""GPU tests of the all-pairs similarity + selection kernels (include/n2v_sim.h),
through the C-ABI, against the plain-Python restatements of src/main_link.py:62-170 and :351-475 under oracle/.
PARITY UNPINNED for both rows (main_link.py does not import here and holds no fixture): the comparison is with
the restated text.  Scores are fp32 on both sides but summed in a different order: exact SETS are required
except where two scores are closer than 2e-6 at the cut."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("sim_method", ["cos", "pearson"])
@pytest.mark.parametrize("mode,ratio,thre", [("ratio", 0.1, 0.5), ("ratio", 0.37, 0.5), ("step", 0.1, 0.1),
                                             ("relu", 0.1, 0.05), ("relu-ratio", 0.12, 0.9), ("linear", 0.1, 0.5)])
def test_selection_matches_restatement(torch_cuda, mode, ratio, thre, sim_method):
    torch = torch_cuda
    from n2v_hip import augment
    from oracle import augment_oracle
    rs = np.random.RandomState(5)
    n, d = 157, 48
    vec = rs.normal(size=(n, d)).astype(np.float32)
    users = [int(x) for x in rs.permutation(1000)[:n]]
    emb = {u: vec[i] for i, u in enumerate(users)}
    want = augment_oracle.add_user_edge(users, emb, mode, ratio, thre, sim_method)
    s, t, w = augment.add_edges(torch.from_numpy(vec).cuda(), mode, ratio, thre, block_rows=50, sim_method=sim_method)
    got = [(users[a], users[b], float(c)) for a, b, c in zip(s.tolist(), t.tolist(), w.tolist())]
    assert len(got) == len(want)
    assert [(a, b) for a, b, _ in got] == [(a, b) for a, b, _ in want]
    np.testing.assert_allclose([c for _, _, c in got], [float(c) for _, _, c in want], atol=3e-6)


def test_jsd_scores_match_restatement(torch_cuda):
    """js() of src/main_link.py:351-356 on vectors with positive entries (finite) and with a negative entry
    (scipy's rel_entr is +inf there, and so is the kernel's)."""
    torch = torch_cuda
    from n2v_hip import simsel
    from oracle import augment_oracle
    rs = np.random.RandomState(3)
    vec = rs.random_sample((40, 20)).astype(np.float32) + 0.01
    vec[7, 3] = -0.2
    X = simsel.prepare(torch.from_numpy(vec).cuda(), "jsd")
    S = simsel.score_block(X, 0, 40, X, "jsd").cpu().numpy()
    for i, j in [(0, 1), (5, 30), (12, 12), (39, 2)]:
        want = augment_oracle.js(vec[i].astype(np.float64), vec[j].astype(np.float64))
        assert abs(S[i, j] - want) < 2e-6, (i, j, S[i, j], want)
    assert np.isinf(S[7, :]).all() and np.isinf(S[:, 7]).all()
    assert np.isinf(augment_oracle.js(vec[7].astype(np.float64), vec[1].astype(np.float64)))


def test_rows_topk_with_ties_is_the_stable_sort_prefix(torch_cuda):
    """sorted(zip(nodes, sims), key=-sim)[:k] keeps list order among equal scores (:391-393): the radix select
    must return exactly that prefix, for every k, on rows full of ties, negatives, zeros and a NaN."""
    torch = torch_cuda
    from n2v_hip import simsel
    rs = np.random.RandomState(0)
    n_rows, n_cols = 9, 1000
    sc = rs.randint(-3, 4, size=(n_rows, n_cols)).astype(np.float32) / 4
    sc[1] = 0.25
    sc[2, ::3] = -0.0
    sc[3, 17] = np.nan
    sc[4] = rs.normal(size=n_cols).astype(np.float32)
    dev = torch.from_numpy(sc).cuda()
    for k in (1, 2, 255, 256, 257, 999, 1000):
        cols, vals = simsel.rows_topk(dev, n_cols, k)
        cols, vals = cols.cpu().numpy(), vals.cpu().numpy()
        for r in range(n_rows):
            key = np.where(np.isnan(sc[r]), -np.inf, sc[r])
            want = np.argsort(-key, kind="stable")[:k]
            if np.isnan(sc[r]).any() and k == n_cols:
                assert set(cols[r].tolist()) == set(want.tolist())
                continue
            assert cols[r].tolist() == want.tolist(), (k, r)
            assert np.array_equal(vals[r], sc[r][want])


def test_rows_above_matches_numpy(torch_cuda):
    torch = torch_cuda
    from n2v_hip import simsel
    rs = np.random.RandomState(1)
    sc = rs.normal(size=(33, 777)).astype(np.float32)
    dev = torch.from_numpy(sc).cuda()
    for thre in (-10.0, 0.0, 0.5, 10.0):
        r, c, v = simsel.rows_above(dev, 777, thre)
        wr, wc = np.nonzero(sc > thre)
        assert np.array_equal(r.cpu().numpy(), wr) and np.array_equal(c.cpu().numpy(), wc)
        assert np.array_equal(v.cpu().numpy(), sc[wr, wc])


def _bipartite_case(n_u=1200, n_i=800, d=128, seed=0):
    """Users 0..n_u-1, items 9999999<id>; embedding with community structure so the top scores are spread."""
    rs = np.random.RandomState(seed)
    comm_u, comm_i = rs.randint(0, 20, n_u), rs.randint(0, 20, n_i)
    centres = rs.normal(size=(20, d))
    vu = centres[comm_u] + 0.7 * rs.normal(size=(n_u, d))
    vi = centres[comm_i] + 0.7 * rs.normal(size=(n_i, d))
    users = np.arange(n_u, dtype=np.int64)
    items = np.array([int("9999999%d" % i) for i in range(n_i)], dtype=np.int64)
    m = 12000
    e = np.stack([users[rs.randint(0, n_u, m)], items[rs.randint(0, n_i, m)]], 1)
    e = np.unique(e, axis=0)
    rs.shuffle(e)
    return users, items, vu.astype(np.float32), vi.astype(np.float32), e[: len(e) // 2], e[len(e) // 2:]


def _check_topk(results, final, want_results, want_final, ks):
    for k in ks:
        got, want = results[k], want_results[k]
        assert len(got) == len(want) == min(k, len(want))
        gs, ws = {p for p, _, _ in got}, {p for p, _, _ in want}
        if gs != ws:
            # only pairs tied with the k-th score (to fp32 summation noise) may differ
            cut = want[-1][1]
            for p, s, _ in got + want:
                if p in gs ^ ws:
                    assert abs(s - cut) < 2e-6, (k, p, s, cut)
        np.testing.assert_allclose(sorted(s for _, s, _ in got), sorted(s for _, s, _ in want), atol=2e-6)
        if gs == ws:
            assert final[k] == pytest.approx(want_final[k]), k
            pops = {p: q for p, _, q in want}
            assert all(pops[p] == q for p, _, q in got)


def test_link_prediction_topk_matches_restatement(torch_cuda):
    """src/main_link.py:123-170 on a 2 000-node bipartite graph: the device's running top-k over 1200 x 800
    pairs minus the training edges == the restatement's, with precision and average popularity."""
    torch = torch_cuda
    from n2v_hip import csr, linkpred
    from oracle import linkpred_oracle as lo
    users, items, vu, vi, train, test = _bipartite_case()
    g = csr.from_edges(train[:, 0], train[:, 1], None, False)
    # every node must be in the graph's label set: add the isolated ones through the full graph
    full = csr.from_edges(np.concatenate([train[:, 0], users, items[:-1]]), np.concatenate([train[:, 1], users, items[1:]]), None, False)
    g = linkpred._with_isolated_nodes(g, full)
    vec = np.zeros((g.n_nodes, 128), dtype=np.float32)
    vec[g.dense_of(users)] = vu
    vec[g.dense_of(items)] = vi
    res, fin = linkpred.link_prediction(torch.from_numpy(vec).cuda(), g, train, test)
    emb = {str(int(l)): vec[i] for i, l in enumerate(g.labels)}
    adj = {int(l): range(int(g.row_ptr[i + 1] - g.row_ptr[i])) for i, l in enumerate(g.labels)}
    wres, wfin = lo.link_prediction_vectorised(False, adj, emb, train.tolist(), test.tolist())
    _check_topk(res, fin, wres, wfin, lo.KS)
    assert fin[1000][0] >= 0.0 and len(res[1000]) == 1000
    # no training edge among the predictions
    tr = {(str(a), str(b)) for a, b in train.tolist()}
    assert not any(p in tr for p, _, _ in res[1000])


def test_link_prediction_unseparated_and_small_buffer(torch_cuda):
    """`unseparated`: pairs nodes[i], nodes[j], i < j (:72).  Also forces the candidate buffer to overflow
    (capacity 2 048 < candidates of the first row blocks) so the raise-threshold-and-rescan path runs."""
    torch = torch_cuda
    from n2v_hip import csr, linkpred, simsel
    from oracle import linkpred_oracle as lo
    rs = np.random.RandomState(4)
    n, d = 900, 64
    labels = np.sort(rs.permutation(5000)[:n]).astype(np.int64)
    vec = (rs.normal(size=(20, d))[rs.randint(0, 20, n)] + 0.5 * rs.normal(size=(n, d))).astype(np.float32)
    e = labels[rs.randint(0, n, size=(6000, 2))]
    e = e[e[:, 0] != e[:, 1]]
    train, test = e[:3000], e[3000:]
    g = csr.from_edges(np.concatenate([train[:, 0], labels[:-1]]), np.concatenate([train[:, 1], labels[1:]]), None, False)
    dv = torch.from_numpy(vec).cuda()
    res, fin = linkpred.link_prediction(dv, g, train, test, unseparated=True)
    emb = {str(int(l)): vec[i] for i, l in enumerate(g.labels)}
    adj = {int(l): range(int(g.row_ptr[i + 1] - g.row_ptr[i])) for i, l in enumerate(g.labels)}
    wres, wfin = lo.link_prediction_vectorised(True, adj, emb, train.tolist(), test.tolist())
    _check_topk(res, fin, wres, wfin, lo.KS)
    A = simsel.prepare(dv, "cos")
    s1, r1, c1 = simsel.global_topk(A, A, 1000, upper_triangle=True)
    s2, r2, c2 = simsel.global_topk(A, A, 1000, upper_triangle=True, capacity=2048, first_rows=256)
    assert torch.equal(s1, s2) and torch.equal(r1, r2) and torch.equal(c1, c2)
    assert bool((c1 > r1).all())
