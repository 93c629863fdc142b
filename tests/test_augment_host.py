"""CPU tests of the edge-augmentation row: the host-side graph surgery —
add_weighted_edges_from semantics against networkx itself, the user-node filter.  The similarity + selection
kernels are checked against the restatement of src/main_link.py:351-475 in tests/test_gpu_sim.py."""
import numpy as np
import pytest
import torch

from n2v_hip import augment, csr


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
