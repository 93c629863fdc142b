"""Pin the oracle (oracle/n2v_oracle.py and oracle/n2v_oracle.c) to the golden vectors
captured from the reference's src/node2vec.py by tests/golden/make_golden.py."""
import numpy as np
import pytest

from helpers import GRAPH_CASES, golden_walks, load_case, oracle_graph, walks_from_padded
from oracle import c_oracle
from oracle import n2v_oracle as orc


def _bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


def test_alias_setup_known_answers(golden_dir):
    z = dict(np.load(golden_dir + "/alias_setup.npz"))
    ptr = z["ptr"]
    for t in range(len(ptr) - 1):
        pr = z["probs"][ptr[t]:ptr[t + 1]]
        J, q = orc.alias_setup([float(x) for x in pr])
        assert np.array_equal(J, z["J"][ptr[t]:ptr[t + 1]])
        assert np.array_equal(_bits(q), _bits(z["q"][ptr[t]:ptr[t + 1]]))
        Jc, qc = c_oracle.alias_setup(pr)
        assert np.array_equal(Jc, J)
        assert np.array_equal(_bits(qc), _bits(q))


def test_alias_setup_uniform_quirk(golden_dir):
    # K*(1.0/K) == 0.9999999999999999 for K = 49: all slots land in `smaller`, J stays 0
    J, q = orc.alias_setup([1.0 / 49] * 49)
    assert (J == 0).all() and (q < 1.0).all()


def test_alias_draw_known_answers(golden_dir):
    z = dict(np.load(golden_dir + "/alias_setup.npz"))
    t = int(z["draw_table"])
    ptr = z["ptr"]
    J, q = z["J"][ptr[t]:ptr[t + 1]], z["q"][ptr[t]:ptr[t + 1]]
    rs = np.random.RandomState(int(z["draw_seed"]))
    got = [orc.alias_draw_u(J, q, rs.random_sample(), rs.random_sample()) for _ in range(len(z["draws"]))]
    assert got == z["draws"].tolist()


def test_mt19937_matches_numpy():
    for seed in (0, 1, 123, 2**32 - 1):
        a = c_oracle.mt19937_fill(seed, 5000)
        b = np.random.RandomState(seed).random_sample(5000)
        assert np.array_equal(_bits(a), _bits(b))
    a = c_oracle.mt19937_fill(123, 100, skip=777)
    b = np.random.RandomState(123).random_sample(877)[777:]
    assert np.array_equal(_bits(a), _bits(b))


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
        ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
        ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
         (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
    ]
    for ctr, key, want in kat:
        assert orc.philox4x32_10(ctr, key) == want
        assert c_oracle.philox4x32_10(ctr, key) == want


@pytest.mark.parametrize("name", GRAPH_CASES)
def test_python_oracle_tables_and_walks(name):
    z = load_case(name)
    G = oracle_graph(z)
    assert G.nodes == z["nodes"].tolist()
    o = orc.Node2VecOracle(G, bool(z["directed"]), float(z["p"]), float(z["q"]))
    big = len(z["ae_J"]) > 20000
    if not big:
        o.preprocess_transition_probs()
        # sorted adjacency + node tables
        ap = z["adj_ptr"]
        for i, v in enumerate(G.nodes):
            assert sorted(G.neighbors(v)) == z["adj"][ap[i]:ap[i + 1]].tolist()
            J, q = o.alias_nodes[v]
            assert np.array_equal(J, z["an_J"][ap[i]:ap[i + 1]])
            assert np.array_equal(_bits(q), _bits(z["an_q"][ap[i]:ap[i + 1]]))
        # edge tables: same key order, same bits
        assert [list(k) for k in o.alias_edges.keys()] == z["ae_keys"].tolist()
        ep = z["ae_ptr"]
        for i, k in enumerate(o.alias_edges.keys()):
            J, q = o.alias_edges[k]
            assert np.array_equal(J, z["ae_J"][ep[i]:ep[i + 1]])
            assert np.array_equal(_bits(q), _bits(z["ae_q"][ep[i]:ep[i + 1]]))
    for i, (seed, r, L, ndraws, has_sub, fly) in enumerate(z["walk_meta"].tolist()):
        if big and r * L > 100:
            continue
        sub = z["walks_%d_subset" % i].tolist() if has_sub else None
        rs = np.random.RandomState(seed)
        walks = o.simulate_walks(r, L, nodes=sub, rand=rs.random_sample, on_the_fly=bool(fly or big))
        assert walks == golden_walks(z, i)
        # consumed exactly ndraws uniforms: the next double of a fresh stream agrees
        chk = np.random.RandomState(seed)
        chk.random_sample(ndraws)
        assert chk.random_sample() == rs.random_sample()


@pytest.mark.parametrize("name", GRAPH_CASES)
def test_c_oracle_tables_and_walks(name):
    z = load_case(name)
    G = oracle_graph(z)
    labels, row_ptr, col, w, start_order = orc.to_csr(G)
    rank = {int(l): i for i, l in enumerate(labels)}
    co = c_oracle.CsrOracle(row_ptr, col, w, float(z["p"]), float(z["q"]))
    co.preprocess()
    # node tables in fixture (insertion) order vs CSR (label-rank) order
    ap = z["adj_ptr"]
    for i, v in enumerate(z["nodes"].tolist()):
        d = rank[v]
        sl = slice(row_ptr[d], row_ptr[d + 1])
        assert np.array_equal(labels[col[sl]], z["adj"][ap[i]:ap[i + 1]])
        assert np.array_equal(co.nodeJ[sl], z["an_J"][ap[i]:ap[i + 1]])
        assert np.array_equal(_bits(co.nodeq[sl]), _bits(z["an_q"][ap[i]:ap[i + 1]]))
    ep = z["ae_ptr"]
    assert len(z["ae_keys"]) == co.nnz  # one table per CSR entry, nothing more
    for i, (u, v) in enumerate(z["ae_keys"].tolist()):
        du, dv = rank[u], rank[v]
        e = row_ptr[du] + int(np.searchsorted(col[row_ptr[du]:row_ptr[du + 1]], dv))
        assert col[e] == dv
        sl = slice(co.edge_off[e], co.edge_off[e + 1])
        assert np.array_equal(co.edgeJ[sl], z["ae_J"][ep[i]:ep[i + 1]])
        assert np.array_equal(_bits(co.edgeq[sl]), _bits(z["ae_q"][ep[i]:ep[i + 1]]))
    for i, (seed, r, L, ndraws, has_sub, fly) in enumerate(z["walk_meta"].tolist()):
        starts = ([rank[x] for x in z["walks_%d_subset" % i].tolist()] if has_sub else start_order)
        for otf in (False, True):
            walks, lens, n = co.walk(starts, r, L, mode="mt", seed=seed, on_the_fly=otf)
            assert n == ndraws
            assert walks_from_padded(walks, lens, labels) == golden_walks(z, i)
        # buffer mode == sequential MT stream
        U = np.random.RandomState(seed).random_sample(max(ndraws, 1))
        walks, lens, n = co.walk(starts, r, L, mode="buffer", uniforms=U)
        assert n == ndraws and walks_from_padded(walks, lens, labels) == golden_walks(z, i)


def test_first_order_shortcut_equals_edge_tables():
    # p == q == 1: every (src,dst) table equals dst's node table (bit for bit)
    z = load_case("karate_p1_q1")
    G = oracle_graph(z)
    labels, row_ptr, col, w, start_order = orc.to_csr(G)
    co = c_oracle.CsrOracle(row_ptr, col, w, 1.0, 1.0)
    co.preprocess()
    for e in range(co.nnz):
        d = col[e]
        sl = slice(co.edge_off[e], co.edge_off[e + 1])
        assert np.array_equal(co.edgeJ[sl], co.nodeJ[row_ptr[d]:row_ptr[d + 1]])
        assert np.array_equal(_bits(co.edgeq[sl]), _bits(co.nodeq[row_ptr[d]:row_ptr[d + 1]]))


def test_philox_walks_python_vs_c():
    z = load_case("karate_p025_q4")
    G = oracle_graph(z)
    o = orc.Node2VecOracle(G, False, 0.25, 4.0)
    o.preprocess_transition_probs()
    labels, row_ptr, col, w, start_order = orc.to_csr(G)
    co = c_oracle.CsrOracle(row_ptr, col, w, 0.25, 4.0)
    co.preprocess()
    seed = 0x1234567890AB
    py = o.simulate_walks(2, 12, step_uniforms=lambda wi, t: orc.philox_step_uniforms(seed, wi + 1000, t))
    walks, lens, _ = co.walk(start_order, 2, 12, mode="philox", seed=seed, walk_index_base=1000)
    assert walks_from_padded(walks, lens, labels) == py


def test_zero_weight_raises():
    from oracle.n2v_oracle import OracleGraph
    G = OracleGraph([(0, 1), (1, 2)], [0.0, 0.0], False)
    o = orc.Node2VecOracle(G, False, 1.0, 1.0)
    with pytest.raises(ZeroDivisionError):
        o.preprocess_transition_probs()
    labels, row_ptr, col, w, _ = orc.to_csr(G)
    with pytest.raises(ZeroDivisionError):
        c_oracle.CsrOracle(row_ptr, col, w, 1.0, 1.0).preprocess()
