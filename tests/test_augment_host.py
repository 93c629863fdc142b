"""CPU tests of the edge-augmentation row: the device selection code (run on
CPU tensors here) against the oracle's restatement of src/main_link.py:379-475, and the
add_weighted_edges_from semantics against networkx itself."""
import numpy as np
import pytest
import torch

from n2v_hip import augment, csr
from oracle import augment_oracle


@pytest.mark.parametrize("mode,ratio,thre", [("ratio", 0.1, 0.5), ("ratio", 0.37, 0.5), ("step", 0.1, 0.1),
                                             ("relu", 0.1, 0.05), ("relu-ratio", 0.12, 0.9), ("linear", 0.1, 0.5)])
def test_selection_matches_oracle(mode, ratio, thre):
    rs = np.random.RandomState(5)
    n, d = 57, 16
    vec = rs.normal(size=(n, d)).astype(np.float32)
    users = [int(x) for x in rs.permutation(1000)[:n]]
    emb = {u: vec[i] for i, u in enumerate(users)}
    want = augment_oracle.add_user_edge(users, emb, mode, ratio, thre)
    s, t, w = augment.add_edges(torch.from_numpy(vec), mode, ratio, thre, block_rows=16)
    got = [(users[a], users[b], float(c)) for a, b, c in zip(s.tolist(), t.tolist(), w.tolist())]
    assert len(got) == len(want)
    assert [(a, b) for a, b, _ in got] == [(a, b) for a, b, _ in want]
    np.testing.assert_allclose([c for _, _, c in got], [float(c) for _, _, c in want], atol=2e-6)


def test_user_nodes_filter():
    labels = np.array([5, 99999991, 12, 9999999, 99999990001, 999999])
    assert augment.user_nodes(labels).tolist() == [5, 12, 999999]
    assert augment.user_nodes(labels, unseparated=True).tolist() == labels.tolist()


@pytest.mark.parametrize("directed", [False, True])
def test_add_weighted_edges_matches_networkx(directed):
    import networkx as nx
    rs = np.random.RandomState(2)
    for trial in range(10):
        n, m = 12, 30
        src, dst = rs.randint(0, n, m), rs.randint(0, n, m)
        g0 = csr.from_edges(src, dst, None, directed)
        nodes = g0.labels
        k = 25
        a, b = nodes[rs.randint(0, len(nodes), k)], nodes[rs.randint(0, len(nodes), k)]
        w = rs.randint(1, 9, k) / 4.0
        g1 = augment.add_weighted_edges(g0, a, b, w)
        G = nx.DiGraph()
        for u, v in zip(src.tolist(), dst.tolist()):
            G.add_edge(u, v, weight=1)
        if not directed:
            G = G.to_undirected()
        G.add_weighted_edges_from(zip(a.tolist(), b.tolist(), w.tolist()))
        ref = csr.from_networkx(G)
        assert np.array_equal(g1.row_ptr, ref.row_ptr) and np.array_equal(g1.col, ref.col)
        assert np.array_equal(g1.w, np.ones(ref.nnz) if ref.w is None else ref.w)
        assert np.array_equal(g1.labels, ref.labels)
