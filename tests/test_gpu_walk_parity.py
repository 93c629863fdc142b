"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the
C-ABI and the drop-in ``node2vec`` module, against (1) the golden vectors captured from the
reference and (2) the oracle on seeded inputs.  Bar: bit-exact (tables: int J and the raw
bits of fp64 q; walks: identical node sequences)."""
import numpy as np
import pytest

from helpers import GRAPH_CASES, case_weights, golden_walks, load_case, oracle_graph, walks_from_padded

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _nx_graph(z):
    import networkx as nx
    G = nx.DiGraph()
    for (u, v), w in zip(z["edges"].tolist(), case_weights(z)):
        G.add_edge(int(u), int(v), weight=w)
    if not bool(z["directed"]):
        G = G.to_undirected()
    return G


@pytest.fixture(scope="module")
def n2v():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import node2vec
    return node2vec


def test_alias_setup_known_answers_gpu(n2v, golden_dir):
    z = dict(np.load(golden_dir + "/alias_setup.npz"))
    ptr = z["ptr"]
    from n2v_hip.engine import alias_setup_device
    tables = [z["probs"][ptr[t]:ptr[t + 1]].tolist() for t in range(len(ptr) - 1)]
    nonempty = [t for t in tables if len(t)]
    res = alias_setup_device(nonempty)
    i = 0
    for t, pr in enumerate(tables):
        if not len(pr):
            J, q = n2v.alias_setup([])
            assert len(J) == 0 and len(q) == 0
            continue
        J, q = res[i]
        i += 1
        assert np.array_equal(J, z["J"][ptr[t]:ptr[t + 1]]), t
        assert np.array_equal(_bits(q), _bits(z["q"][ptr[t]:ptr[t + 1]])), t
    # single-table module-level entry point (src/node2vec.py:240)
    J, q = n2v.alias_setup(tables[140])
    assert np.array_equal(J, z["J"][ptr[140]:ptr[141]]) and J.dtype == np.int64
    assert np.array_equal(_bits(q), _bits(z["q"][ptr[140]:ptr[141]]))


@pytest.mark.parametrize("name", GRAPH_CASES)
def test_tables_match_reference(n2v, name):
    z = load_case(name)
    g = n2v.Graph(_nx_graph(z), bool(z["directed"]), float(z["p"]), float(z["q"]))
    assert g.preprocess_transition_probs() is None
    eng = g._engine
    if eng.first_order:  # p == q == 1 shares tables; also check the materialised form
        eng.preprocess(first_order_shortcut=False)
    ap = z["adj_ptr"]
    nodes = z["nodes"].tolist()
    assert g.alias_nodes.keys() == nodes
    nq = eng.slots_q(eng.node_slots).cpu().numpy()
    nJ = eng.slots_J(eng.node_slots).cpu().numpy()
    csr = g._csr
    for i, v in enumerate(nodes):
        d = int(csr.dense_of([v])[0])
        sl = slice(int(csr.row_ptr[d]), int(csr.row_ptr[d + 1]))
        assert np.array_equal(csr.labels[csr.col[sl]], z["adj"][ap[i]:ap[i + 1]])
        assert np.array_equal(nJ[sl], z["an_J"][ap[i]:ap[i + 1]])
        assert np.array_equal(_bits(nq[sl]), _bits(z["an_q"][ap[i]:ap[i + 1]]))
    J0, q0 = g.alias_nodes[nodes[0]]
    assert J0.dtype == np.int64 and np.array_equal(J0, z["an_J"][ap[0]:ap[1]])

    eJ, eq = eng.all_edge_tables()      # decoded from the layout the engine stores (fat slots by default)
    eoff = eng.edge_off.cpu().numpy()
    ep = z["ae_ptr"]
    assert len(g.alias_edges) == len(z["ae_keys"])
    assert sorted(g.alias_edges.keys()) == sorted(map(tuple, z["ae_keys"].tolist()))
    for i, (u, v) in enumerate(z["ae_keys"].tolist()):
        e = eng.edge_index(int(csr.dense_of([u])[0]), int(csr.dense_of([v])[0]))
        assert e >= 0
        sl = slice(eoff[e], eoff[e + 1])
        assert np.array_equal(eJ[sl], z["ae_J"][ep[i]:ep[i + 1]]), (u, v)
        assert np.array_equal(_bits(eq[sl]), _bits(z["ae_q"][ep[i]:ep[i + 1]])), (u, v)
    k = tuple(z["ae_keys"][0].tolist())
    Jk, qk = g.alias_edges[k]
    assert np.array_equal(Jk, z["ae_J"][ep[0]:ep[1]]) and np.array_equal(_bits(qk), _bits(z["ae_q"][ep[0]:ep[1]]))
    assert (10**15, 1) not in g.alias_edges
    with pytest.raises(KeyError):
        g.alias_edges[(nodes[0], 10**15)]
    # get_alias_edge / get_alias_edges_cur / get_alias_nodes_cur (src/node2vec.py:13-32,133-152): one table built
    # on demand, on a graph object that never ran preprocess_transition_probs
    g2 = n2v.Graph(_nx_graph(z), bool(z["directed"]), float(z["p"]), float(z["q"]))
    for i in list(range(0, len(z["ae_keys"]), max(1, len(z["ae_keys"]) // 25)))[:40]:
        u, v = z["ae_keys"][i].tolist()
        for fn in (g2.get_alias_edge, g2.get_alias_edges_cur):
            J, q = fn(u, v)
            assert np.array_equal(J, z["ae_J"][ep[i]:ep[i + 1]]), (u, v)
            assert np.array_equal(_bits(q), _bits(z["ae_q"][ep[i]:ep[i + 1]])), (u, v)
    for i, v in list(enumerate(nodes))[:: max(1, len(nodes) // 20)]:
        J, q = g2.get_alias_nodes_cur(v)
        assert np.array_equal(J, z["an_J"][ap[i]:ap[i + 1]]) and np.array_equal(_bits(q), _bits(z["an_q"][ap[i]:ap[i + 1]]))
    with pytest.raises(KeyError):
        g2.get_alias_edge(nodes[0], 10**15)


@pytest.mark.parametrize("name", GRAPH_CASES)
def test_walks_match_reference_under_numpy_seed(n2v, name):
    """np.random.seed(s); G.simulate_walks(...) == the reference's output, and the global
    MT19937 stream is left exactly where the reference leaves it."""
    z = load_case(name)
    g = n2v.Graph(_nx_graph(z), bool(z["directed"]), float(z["p"]), float(z["q"]))
    g.preprocess_transition_probs()
    for i, (seed, r, L, ndraws, has_sub, fly) in enumerate(z["walk_meta"].tolist()):
        sub = z["walks_%d_subset" % i].tolist() if has_sub else None
        np.random.seed(seed)
        fn = g.simulate_walks_on_the_fly if fly else g.simulate_walks
        walks = fn(r, L, nodes=sub)
        want = golden_walks(z, i)
        assert len(walks) == len(want)
        assert walks == want, (name, i)
        chk = np.random.RandomState(seed)
        chk.random_sample(ndraws)
        assert np.random.random_sample() == chk.random_sample(), (name, i, "global stream position")
    # node2vec_walk (src/node2vec.py:55-79), used directly by the reference's pool workers
    np.random.seed(77)
    start = z["nodes"].tolist()[0]
    w1 = g.node2vec_walk(12, start)
    from oracle import n2v_oracle as orc
    o = orc.Node2VecOracle(oracle_graph(z), bool(z["directed"]), float(z["p"]), float(z["q"]))
    rs = np.random.RandomState(77)
    assert w1 == o.node2vec_walk(12, start, rs.random_sample, on_the_fly=True)


def test_unknown_start_node_raises(n2v):
    z = load_case("karate_p1_q1")
    g = n2v.Graph(_nx_graph(z), False, 1, 1)
    g.preprocess_transition_probs()
    with pytest.raises(KeyError):
        g.simulate_walks(1, 5, nodes=[1, 999])


def test_zero_weight_raises_zero_division(n2v):
    import networkx as nx
    G = nx.Graph()
    G.add_edge(0, 1, weight=0.0)
    G.add_edge(1, 2, weight=0.0)
    g = n2v.Graph(G, False, 1, 1)
    with pytest.raises(ZeroDivisionError):
        g.preprocess_transition_probs()
    g = n2v.Graph(G, False, 0.5, 2)
    with pytest.raises(ZeroDivisionError):
        g.preprocess_transition_probs()


def _random_graph(n, m, seed, weighted, directed):
    rs = np.random.RandomState(seed)
    src = rs.randint(0, n, size=m)
    dst = rs.randint(0, n, size=m)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    w = (rs.random_sample(len(src)) * 3 + 0.25) if weighted else None
    from n2v_hip import csr
    return csr.from_edges(src, dst, w, directed)


@pytest.mark.parametrize("weighted,directed,p,q", [(False, False, 0.25, 4.0), (True, False, 2.0, 0.5),
                                                     (True, True, 0.5, 2.0), (False, False, 1.0, 1.0)])
def test_tables_and_walks_vs_c_oracle_20k(n2v, weighted, directed, p, q):
    """20k nodes / ~100k edges: every slot of every table and every walk vs the C oracle,
    in parity mode (uniform buffer) and in throughput mode (Philox)."""
    import torch
    from oracle import c_oracle
    cg = _random_graph(20000, 100000, 5, weighted, directed)
    g = n2v.Graph.from_csr(cg, p, q, rng="philox", seed=0xC0FFEE1234)
    g.preprocess_transition_probs()
    eng = g._engine
    co = c_oracle.CsrOracle(cg.row_ptr, cg.col, cg.w, p, q)
    co.preprocess(first_order_shortcut=eng.first_order)
    assert np.array_equal(eng.slots_J(eng.node_slots).cpu().numpy()[:cg.nnz], co.nodeJ)
    assert np.array_equal(_bits(eng.slots_q(eng.node_slots).cpu().numpy()[:cg.nnz]), _bits(co.nodeq))
    if not eng.first_order:
        assert np.array_equal(eng.edge_off.cpu().numpy(), co.edge_off)
        T = int(co.edge_off[-1])
        assert eng.edge_slots is None and eng.edge_fat is not None    # default: fat tables only, no thin copy
        eJ, eq = eng.all_edge_tables()
        assert np.array_equal(eJ[:T], co.edgeJ) and np.array_equal(_bits(eq[:T]), _bits(co.edgeq))
        # thin output of the wave kernel and round 1's lane-per-table kernel: the same bits
        for builder in ("wave", "lane"):
            eng.preprocess(fat="both", builder=builder)
            assert np.array_equal(eng.slots_J(eng.edge_slots).cpu().numpy()[:T], co.edgeJ), builder
            assert np.array_equal(_bits(eng.slots_q(eng.edge_slots).cpu().numpy()[:T]), _bits(co.edgeq)), builder
    else:
        eng.preprocess(fat="both")
    L, r = 40, 2
    # throughput mode
    walks = g.simulate_walks(r, L)
    ow, ol, _ = co.walk(cg.start_order, r, L, mode="philox", seed=0xC0FFEE1234)
    assert np.array_equal(walks.lens.cpu().numpy(), ol)
    assert np.array_equal(walks.walks.cpu().numpy(), ow)
    # parity mode, sequential stream (directed: ragged walks, offsets found by fixed point)
    g.rng = "numpy"
    sub = cg.labels[cg.start_order[:3000]].tolist() if directed else None
    np.random.seed(42)
    walks = g.simulate_walks(1, L, nodes=sub)
    starts = cg.start_order[:3000] if directed else cg.start_order
    ow, ol, nd = co.walk(starts, 1, L, mode="mt", seed=42)
    assert np.array_equal(walks.lens.cpu().numpy(), ol)
    assert np.array_equal(walks.walks.cpu().numpy(), ow)
    chk = np.random.RandomState(42)
    chk.random_sample(nd)
    assert np.random.random_sample() == chk.random_sample()
    # both table layouts (16-B slots + records, 32-B fat slots) give the same walks
    assert eng.edge_fat is not None
    tw, tl = eng.walk(eng.start_order, 2, L, rng="philox", seed=0xC0FFEE1234, layout="thin")
    fw, fl = eng.walk(eng.start_order, 2, L, rng="philox", seed=0xC0FFEE1234, layout="fat")
    ow2, ol2, _ = co.walk(cg.start_order, 2, L, mode="philox", seed=0xC0FFEE1234)
    assert torch.equal(tw, fw) and torch.equal(tl, fl)
    assert np.array_equal(tw.cpu().numpy(), ow2) and np.array_equal(tl.cpu().numpy(), ol2)
    U = torch.rand(2 * (L - 1) * cg.n_nodes, dtype=torch.float64, device=eng.device)
    tw, tl = eng.walk(eng.start_order, 1, L, rng="uniforms", uniforms=U, layout="thin")
    fw, fl = eng.walk(eng.start_order, 1, L, rng="uniforms", uniforms=U, layout="fat")
    assert torch.equal(tw, fw) and torch.equal(tl, fl)
    # sharding by start position and by round reproduces the same rows
    g.rng = "philox"
    full_w, full_l = eng.walk(eng.start_order, 2, L, rng="philox", seed=9)
    n = cg.n_nodes
    a, b = n // 3, n - n // 3
    for rb in (0, 1):
        for (pb, pc) in ((0, a), (a, b)):
            sw, sl = eng.walk(eng.start_order, 1, L, rng="philox", seed=9, pos_begin=pb, pos_count=pc,
                              round_begin=rb)
            assert torch.equal(sw, full_w[rb * n + pb: rb * n + pb + pc])
            assert torch.equal(sl, full_l[rb * n + pb: rb * n + pb + pc])


def test_walk_lengths_not_multiple_of_4_and_length_1(n2v):
    z = load_case("karate_p025_q4")
    g = n2v.Graph(_nx_graph(z), False, 0.25, 4.0)
    g.preprocess_transition_probs()
    from oracle import n2v_oracle as orc
    o = orc.Node2VecOracle(oracle_graph(z), False, 0.25, 4.0)
    o.preprocess_transition_probs()
    for L in (0, 1, 2, 3, 5, 7, 8, 9, 81):
        np.random.seed(3)
        got = g.simulate_walks(2, L)
        want = o.simulate_walks(2, L, seed=3)
        assert got == want, L


@pytest.mark.parametrize("name", GRAPH_CASES)
def test_on_the_fly_kernel_matches_reference(n2v, name):
    """simulate_walks_on_the_fly WITHOUT preprocess_transition_probs (src/node2vec.py:97-111):
    the per-step table rebuild kernel reproduces the reference's walks and stream position."""
    z = load_case(name)
    g = n2v.Graph(_nx_graph(z), bool(z["directed"]), float(z["p"]), float(z["q"]))
    for i, (seed, r, L, ndraws, has_sub, fly) in enumerate(z["walk_meta"].tolist()):
        sub = z["walks_%d_subset" % i].tolist() if has_sub else None
        np.random.seed(seed)
        walks = g.simulate_walks_on_the_fly(r, L, nodes=sub)
        assert g._engine is not None and not g._engine.ready  # no tables were built
        assert walks == golden_walks(z, i), (name, i)
        chk = np.random.RandomState(seed)
        chk.random_sample(ndraws)
        assert np.random.random_sample() == chk.random_sample()
    np.random.seed(5)
    w1 = g.node2vec_walk_on_the_fly(9, z["nodes"].tolist()[0])
    from oracle import n2v_oracle as orc
    o = orc.Node2VecOracle(oracle_graph(z), bool(z["directed"]), float(z["p"]), float(z["q"]))
    assert w1 == o.node2vec_walk(9, z["nodes"].tolist()[0], np.random.RandomState(5).random_sample, on_the_fly=True)


@pytest.mark.parametrize("weighted,directed,p,q", [
    (False, False, 0.25, 4.0),   # dyadic 1/p, 1/q on an unweighted undirected graph: the counting path (n2v_wave_table.h)
    (False, False, 1.0, 1.0),    # dyadic, every class weight 1
    (False, False, 4.0, 0.5),    # dyadic, the return slot small and the far slots large
    (False, False, 0.3, 0.7),    # not dyadic: weights summed left to right, draw before the pairing
    (True, False, 0.25, 4.0),    # weighted: general path
    (False, True, 0.25, 4.0),    # directed: has_edge(nbr, prev) is not a row intersection -> general path
    (True, True, 0.5, 2.0)])
def test_on_the_fly_kernel_equals_table_walk_with_hubs(n2v, weighted, directed, p, q):
    """20k nodes plus hubs of degree 700 and 3000 (tables beyond the 512-slot LDS window go
    through the global scratch path; a hub as `prev` is a row too long for the LDS row cache, a hub as `cur` makes the
    counting path walk prev's row instead): on-the-fly walks == table-driven walks, bit for bit."""
    import torch
    rs = np.random.RandomState(11)
    n, m = 20000, 80000
    src = rs.randint(0, n, size=m)
    dst = rs.randint(0, n, size=m)
    hub_s = np.concatenate([np.full(700, 5), np.full(3000, 17)])
    hub_d = np.concatenate([rs.choice(n, 700, replace=False), rs.choice(n, 3000, replace=False)])
    src, dst = np.concatenate([src, hub_s, hub_d[:500]]), np.concatenate([dst, hub_d, hub_s[:500]])
    keep = src != dst
    src, dst = src[keep], dst[keep]
    w = (rs.random_sample(len(src)) * 3 + 0.25) if weighted else None
    from n2v_hip import csr
    cg = csr.from_edges(src, dst, w, directed)
    assert cg.degrees.max() > 2500
    g = n2v.Graph.from_csr(cg, p, q, rng="philox", seed=77)
    g.preprocess_transition_probs()
    a = g.simulate_walks(1, 30)
    g.force_on_the_fly = True
    b = g.simulate_walks_on_the_fly(1, 30)
    assert torch.equal(a.lens, b.lens) and torch.equal(a.walks, b.walks)
    # parity stream + start subset through the scratch path
    g.rng = "numpy"
    sub = cg.labels[cg.start_order[:500]].tolist() + [5, 17]
    np.random.seed(3)
    c = g.simulate_walks_on_the_fly(2, 12, nodes=sub)
    g.force_on_the_fly = False
    np.random.seed(3)
    d = g.simulate_walks(2, 12, nodes=sub)
    assert torch.equal(c.walks, d.walks) and torch.equal(c.lens, d.lens)


@pytest.mark.parametrize("name", ["hub520_directed", "karate_p025_q4", "er600_p05_q2", "er600_directed", "weighted_toy"])
def test_tables_under_a_memory_budget_walk_like_the_reference(n2v, name):
    """preprocess_transition_probs(budget_bytes=...) with a third of the full tables' size: the tables of entries whose
    destination has the larger degrees are not stored and the walk rebuilds them per step (n2v_walk_hybrid) — the
    reference's walks, the reference's stream position, the reference's tables in the dict views (built on demand
    where they are not stored)."""
    z = load_case(name)
    g = n2v.Graph(_nx_graph(z), bool(z["directed"]), float(z["p"]), float(z["q"]))
    g.preprocess_transition_probs()
    full_bytes = g._engine.total_slots * 32
    if g._engine.first_order:
        pytest.skip("p = q = 1: no edge tables to budget")
    g.preprocess_transition_probs(budget_bytes=full_bytes // 3)
    eng = g._engine
    assert eng.partial and 0 < eng.total_slots * 32 <= full_bytes // 3 and eng.stored_degree_cut < eng.max_degree
    n_unstored = int((~eng.stored_mask).sum())
    assert 0 < n_unstored < eng.csr.nnz
    for i, (seed, r, L, ndraws, has_sub, fly) in enumerate(z["walk_meta"].tolist()):
        if fly:
            continue
        sub = z["walks_%d_subset" % i].tolist() if has_sub else None
        np.random.seed(seed)
        walks = g.simulate_walks(r, L, nodes=sub)
        assert walks == golden_walks(z, i), (name, i)
        chk = np.random.RandomState(seed)
        chk.random_sample(ndraws)
        assert np.random.random_sample() == chk.random_sample(), (name, i, "global stream position")
    # dict views: a stored and an unstored table, both equal to the reference's
    keys = [tuple(k) for k in z["ae_keys"].tolist()]
    ptr = z["ae_ptr"]
    mask = eng.stored_mask.cpu().numpy()
    seen = set()
    for idx, (u, v) in enumerate(keys):
        e = eng.edge_index(int(eng.csr.dense_of([u])[0]), int(eng.csr.dense_of([v])[0]))
        kind = bool(mask[e])
        if kind in seen:
            continue
        seen.add(kind)
        J, q = g.alias_edges[(u, v)]
        assert np.array_equal(J, z["ae_J"][ptr[idx]:ptr[idx + 1]]), (u, v, kind)
        assert np.array_equal(_bits(q), _bits(z["ae_q"][ptr[idx]:ptr[idx + 1]])), (u, v, kind)
        if len(seen) == 2:
            break
    assert len(seen) == 2


def test_tables_under_a_memory_budget_equal_the_c_oracle_with_hubs(n2v):
    """20k nodes plus hubs of degree 700 and 3000, sum of deg^2 = 3x the budget: the stored tables are the low-degree
    destinations' only, the hubs' tables (in LDS up to 512 slots, beyond that in the scratch rows) are rebuilt per
    step — every walk equals the C oracle's (Philox and numpy streams) and the full-table engine's."""
    import torch
    from oracle import c_oracle
    rs = np.random.RandomState(11)
    n, m = 20000, 80000
    src = rs.randint(0, n, size=m)
    dst = rs.randint(0, n, size=m)
    hub_s = np.concatenate([np.full(700, 5), np.full(3000, 17)])
    hub_d = np.concatenate([rs.choice(n, 700, replace=False), rs.choice(n, 3000, replace=False)])
    src, dst = np.concatenate([src, hub_s]), np.concatenate([dst, hub_d])
    keep = src != dst
    from n2v_hip import csr
    cg = csr.from_edges(src[keep], dst[keep], None, False)
    p, q = 0.25, 4.0
    g = n2v.Graph.from_csr(cg, p, q, rng="philox", seed=31)
    g.preprocess_transition_probs()
    full = g._engine.total_slots * 32
    ref = g.simulate_walks(2, 40)
    ref_w, ref_l = ref.walks.clone(), ref.lens.clone()
    g.preprocess_transition_probs(budget_bytes=full // 3)
    eng = g._engine
    assert eng.partial and eng.total_slots * 32 <= full // 3 and eng.stored_degree_cut < eng.max_degree
    assert int((~eng.stored_mask).sum()) > 0
    got = g.simulate_walks(2, 40)
    assert torch.equal(got.walks, ref_w) and torch.equal(got.lens, ref_l)
    co = c_oracle.CsrOracle(cg.row_ptr, cg.col, cg.w, p, q)
    co.preprocess()
    ow, ol, _ = co.walk(cg.start_order, 2, 40, mode="philox", seed=31)
    assert np.array_equal(got.walks.cpu().numpy(), ow) and np.array_equal(got.lens.cpu().numpy(), ol)
    g.rng = "numpy"
    np.random.seed(9)
    got = g.simulate_walks(1, 25)
    ow, ol, _ = co.walk(cg.start_order, 1, 25, mode="mt", seed=9)
    assert np.array_equal(got.walks.cpu().numpy(), ow)
    # no budget given and not even the thin tables fit (simulated: "free" memory of 1.1 GB, 1 GB of it reserve): the tables that fit are kept
    # and the rest is rebuilt per step, instead of a MemoryError
    real = torch.cuda.mem_get_info
    try:
        torch.cuda.mem_get_info = lambda *a, **k: ((1 << 30) + (100 << 20) - (torch.cuda.memory_reserved() - torch.cuda.memory_allocated()), real()[1])
        g.rng = "philox"
        g.preprocess_transition_probs()
    finally:
        torch.cuda.mem_get_info = real
    assert g._engine.partial and g._engine.total_slots == 0      # budget = free - 8 GB < 0: node tables only
    assert torch.equal(g.simulate_walks(2, 40).walks, ref_w)
    # a budget that holds everything changes nothing; one that holds nothing leaves only the node tables
    g.preprocess_transition_probs(budget_bytes=full)
    assert not g._engine.partial
    g.rng = "philox"
    g.preprocess_transition_probs(budget_bytes=0)
    assert g._engine.partial and g._engine.total_slots == 0
    got = g.simulate_walks(2, 40)
    assert torch.equal(got.walks, ref_w)


def test_on_the_fly_zero_weight_raises(n2v):
    import networkx as nx
    G = nx.Graph()
    G.add_edge(0, 1, weight=0.0)
    G.add_edge(1, 2, weight=0.0)
    g = n2v.Graph(G, False, 0.5, 2)
    with pytest.raises(ZeroDivisionError):
        g.simulate_walks_on_the_fly(1, 5)


def test_device_mt19937_equals_numpy_stream(n2v):
    """np.random.random_sample(n) regenerated on the device from numpy's global state: same
    doubles bit for bit, same final global state, from every kind of starting position."""
    import torch
    from n2v_hip import mt19937
    cases = [(123, 0, 10), (123, 0, 1000003), (7, 1001, 312 * 40 + 5), (5, 623, 2_500_001), (9, 1, 1), (9, 0, 311),
             (11, 624 * 3, 624 * 50)]
    for seed, pre_words, n in cases:
        np.random.seed(seed)
        if pre_words:
            np.random.randint(0, 2**32, size=pre_words, dtype=np.uint32)   # leaves an arbitrary (odd) position
        st0 = np.random.get_state()
        want = np.random.random_sample(n)
        st_want = np.random.get_state()
        np.random.set_state(st0)
        got = mt19937.global_uniforms_device(n, "cuda:0")
        st_got = np.random.get_state()
        assert np.array_equal(got.cpu().numpy().view(np.uint64), want.view(np.uint64)), (seed, pre_words, n)
        assert st_got[2] == st_want[2] and np.array_equal(st_got[1], st_want[1]), (seed, pre_words, n)
        # one stream (no jump) gives the same result; so does reading the state back from the kernel
        np.random.set_state(st0)
        got1 = mt19937.global_uniforms_device(n, "cuda:0", n_streams=1, state_from_device=True)
        st_dev = np.random.get_state()
        assert torch.equal(got, got1)
        assert st_dev[2] == st_want[2] and np.array_equal(st_dev[1], st_want[1]), (seed, pre_words, n)
    # start states: the device's doubling scheme equals the host's sequential jumps, for stream counts on
    # both sides of a power of two and for a stride that is not a whole block
    key = np.random.RandomState(77).get_state()[1]
    for stride, ns in ((624 * 211, 37), (624 * 5, 64), (1001, 65), (624, 2), (624 * 3000, 1024)):
        host = mt19937.jump_states(key, stride, min(ns, 80))
        dev = mt19937.jump_states_device(key, stride, ns, "cuda:0").cpu().numpy().view(np.uint32)
        assert np.array_equal(dev[: host.shape[0]], host), (stride, ns)
    np.random.seed(21)
    want = np.random.random_sample(3_000_001)
    st_want = np.random.get_state()
    np.random.seed(21)
    got = mt19937.global_uniforms_device(3_000_001, "cuda:0", n_streams=733)
    assert np.array_equal(got.cpu().numpy().view(np.uint64), want.view(np.uint64))
    assert np.array_equal(np.random.get_state()[1], st_want[1]) and np.random.get_state()[2] == st_want[2]
    # host-side generation stays available and identical
    z = load_case("karate_p025_q4")
    g = n2v.Graph(_nx_graph(z), False, 0.25, 4.0)
    g.preprocess_transition_probs()
    g.host_rng = True
    np.random.seed(123)
    assert g.simulate_walks(10, 80) == golden_walks(z, 0)


def test_degenerate_graphs(n2v):
    """Empty graph, a single self-loop, isolated nodes inside an undirected graph, zero rounds."""
    import networkx as nx
    from n2v_hip import csr
    g = n2v.Graph(nx.Graph(), False, 1, 1)
    g.preprocess_transition_probs()
    assert g.simulate_walks(3, 10) == [] and len(g.alias_nodes) == 0 and len(g.alias_edges) == 0
    G = nx.Graph()
    G.add_edge(7, 7, weight=2.5)
    g = n2v.Graph(G, False, 0.5, 2.0)
    g.preprocess_transition_probs()
    np.random.seed(0)
    assert g.simulate_walks(2, 5) == [[7] * 5, [7] * 5]
    J, q = g.alias_edges[(7, 7)]
    assert J.tolist() == [0] and q.tolist() == [1.0]
    # isolated nodes (degree 0) among connected ones: their walks are [node] and take no draws
    full = csr.CsrGraph(np.arange(8), np.array([0, 2, 4, 6, 6, 6, 7, 8, 8]), np.array([1, 2, 0, 2, 0, 1, 6, 5], dtype=np.int32), None, np.array([3, 0, 7, 1, 2, 4, 5, 6]), False)
    g = n2v.Graph.from_csr(full, 0.5, 2.0, rng="numpy")
    g.preprocess_transition_probs()
    np.random.seed(4)
    walks = g.simulate_walks(2, 6).tolist()
    from oracle import c_oracle
    co = c_oracle.CsrOracle(full.row_ptr, full.col, None, 0.5, 2.0)
    co.preprocess()
    ow, ol, nd = co.walk(full.start_order, 2, 6, mode="mt", seed=4)
    assert walks == walks_from_padded(ow, ol, full.labels)
    assert [w for w in walks if len(w) == 1] == [[3], [7], [4], [3], [7], [4]]
    chk = np.random.RandomState(4)
    chk.random_sample(nd)
    assert np.random.random_sample() == chk.random_sample()
    assert g.simulate_walks(0, 6) == [] and len(g.simulate_walks(0, 6)) == 0


def test_tiled_uniform_layout_is_the_same_stream(n2v):
    """n2v_mt19937_fill_tiled writes exactly the doubles of n2v_mt19937_fill, at the positions the header documents
    (segments of 2(L-1) doubles regrouped 64 at a time, step-major) — for partial groups, segments that straddle
    streams and blocks, odd start positions — and leaves the same global state."""
    import torch
    from n2v_hip import mt19937
    for seed, pre_words, n, pairs in [(3, 0, 158 * 64, 79), (3, 7, 158 * 1000, 79), (4, 301, 2 * 77, 1), (5, 623, 48 * 12345, 24),
                                      (6, 0, 158 * 200003, 79), (8, 11, 6 * 700001, 3), (9, 2, 158 * 5 + 20, 79)]:
        np.random.seed(seed)
        if pre_words:
            np.random.randint(0, 2**32, size=pre_words, dtype=np.uint32)
        st0 = np.random.get_state()
        lin = mt19937.global_uniforms_device(n, "cuda:0")
        st_lin = np.random.get_state()
        np.random.set_state(st0)
        til = mt19937.global_uniforms_device(n, "cuda:0", tiled_pairs=pairs)
        st_til = np.random.get_state()
        assert til.numel() == mt19937.tiled_size(n, pairs)
        idx = torch.from_numpy(mt19937.tiled_index(np.arange(n), pairs)).cuda()
        assert int(idx.max()) < til.numel() and torch.unique(idx).numel() == n
        assert torch.equal(til[idx].view(torch.int64), lin.view(torch.int64)), (seed, pre_words, n, pairs)
        assert st_til[2] == st_lin[2] and np.array_equal(st_til[1], st_lin[1])


def test_tiled_and_linear_uniforms_walk_the_same(n2v):
    """Reference-exact walks read their uniforms from the tiled layout by default (fat tables, no reachable sinks);
    the linear layout and any chunking of the rounds give the same walks and the same
    global state — also with start nodes that have no edges (they own no uniforms) and for walk lengths that take
    the 16-B and the scalar output paths."""
    import torch
    rs = np.random.RandomState(5)
    n, m = 3000, 9000
    src, dst = rs.randint(0, n - 40, m), rs.randint(0, n - 40, m)     # the last 40 ids stay isolated
    keep = src != dst
    import networkx as nx
    G = nx.Graph()
    G.add_nodes_from(rs.permutation(n).tolist())
    G.add_edges_from(zip(src[keep].tolist(), dst[keep].tolist()), weight=1)
    g = n2v.Graph(G, False, 0.5, 2.0)
    g.preprocess_transition_probs()
    assert g._engine.edge_fat is not None
    for L, r in ((80, 7), (36, 3), (7, 2), (2, 3)):
        res = []
        for kw in (dict(linear_uniforms=True), dict(), dict(uniform_chunk_rounds=2), dict(uniform_chunk_rounds=1),
                   dict(uniform_chunk_rounds=3, linear_uniforms=True)):
            for k in ("linear_uniforms", "uniform_chunk_rounds"):
                g.__dict__.pop(k, None)
            g.__dict__.update(kw)
            np.random.seed(77)
            c = g.simulate_walks(r, L)
            st = np.random.get_state()
            res.append((c.walks.clone(), c.lens.clone(), st[1].copy(), st[2]))
        for k in ("linear_uniforms", "uniform_chunk_rounds"):
            g.__dict__.pop(k, None)
        for other in res[1:]:
            assert torch.equal(other[0], res[0][0]) and torch.equal(other[1], res[0][1]), L
            assert np.array_equal(other[2], res[0][2]) and other[3] == res[0][3]
        assert int((res[0][1] == 1).sum()) >= 40 * r      # isolated starts: [node]


def test_shards_reproduce_the_single_process_walks(n2v):
    """Graph.simulate_walks_shard for every rank of a 3-GPU layout, run one after another on this
    GPU: the union equals simulate_walks row for row — in the reference-exact numpy mode (each
    shard jumps the global MT19937 stream to the offsets it owns and leaves the global state
    where the full call would) and in Philox mode."""
    import torch
    z = load_case("er600_p05_q2")
    g = n2v.Graph(_nx_graph(z), False, float(z["p"]), float(z["q"]))
    g.preprocess_transition_probs()
    n, r, L = len(z["nodes"]), 3, 25
    for rng in ("numpy", "philox"):
        g.rng = rng
        g.seed = 99
        np.random.seed(2024)
        full = g.simulate_walks(r, L)
        st_full = np.random.get_state()
        fw = full.walks.view(r, n, L)
        for world in (3, 8):
            for rank in range(world):
                np.random.seed(2024)
                sh = g.simulate_walks_shard(r, L, rank, world)
                per = -(-n // world)
                b, e = min(rank * per, n), min(rank * per + per, n)
                assert torch.equal(sh.walks.view(r, e - b, L), fw[:, b:e]), (rng, world, rank)
                if rng == "numpy":
                    st = np.random.get_state()
                    assert st[2] == st_full[2] and np.array_equal(st[1], st_full[1])
    # the golden walks through the sharded path (world = 1)
    g.rng = "numpy"
    seed, r, L = z["walk_meta"][0][:3].tolist()
    np.random.seed(seed)
    assert g.simulate_walks_shard(r, L, 0, 1) == golden_walks(z, 0)


def test_directed_sinks_sharded_and_pass_bound(n2v):
    """Directed graphs with reachable sinks in the reference-exact mode (SURVEY.md 8(a) row 6'): (1) the golden
    walks of er600_directed through simulate_walks_shard for 3- and 8-rank layouts (round 1 raised
    NotImplementedError there); (2) a 5 000-node graph with 20 sinks (> 10 % of the walks end early): walks equal
    the C oracle's sequential result and the device-side offset resolution needs about one pass (over a WINDOW)
    per walk whose length changes when it is re-walked, ~2 x the share of early-ending walks — round 1 needed up to
    one pass over ALL walks per early-ending walk."""
    import torch
    from n2v_hip import csr
    from oracle import c_oracle
    z = load_case("er600_directed")
    g = n2v.Graph(_nx_graph(z), True, float(z["p"]), float(z["q"]))
    g.preprocess_transition_probs()
    seed, r, L = z["walk_meta"][0][:3].tolist()
    want = golden_walks(z, 0)
    n = len(z["nodes"])
    np.random.seed(seed)
    assert g.simulate_walks(r, L) == want
    st_full = np.random.get_state()
    for world in (3, 8):
        got = [None] * (r * n)
        for rank in range(world):
            np.random.seed(seed)
            sh = g.simulate_walks_shard(r, L, rank, world).tolist()
            st = np.random.get_state()
            assert st[2] == st_full[2] and np.array_equal(st[1], st_full[1])
            per = -(-n // world)
            b, e = min(rank * per, n), min(rank * per + per, n)
            for it in range(r):
                got[it * n + b:it * n + e] = sh[it * (e - b):(it + 1) * (e - b)]
        assert got == want, world
    # many sinks
    rs = np.random.RandomState(8)
    N, M = 5000, 40000
    src, dst = rs.randint(0, N, M), rs.randint(0, N, M)
    keep = (src % 250 != 0) & (src != dst)                 # nodes divisible by 250 have no out-edges: 20 sinks
    cg = csr.from_edges(src[keep], dst[keep], None, True)
    g2 = n2v.Graph.from_csr(cg, 0.5, 2.0, rng="numpy")
    g2.preprocess_transition_probs()
    co = c_oracle.CsrOracle(cg.row_ptr, cg.col, None, 0.5, 2.0)
    co.preprocess()
    r, L = 3, 30
    ow, ol, nd = co.walk(cg.start_order, r, L, mode="mt", seed=77)
    np.random.seed(77)
    got = g2.simulate_walks(r, L)
    assert np.array_equal(got.lens.cpu().numpy(), ol) and np.array_equal(got.walks.cpu().numpy(), ow)
    chk = np.random.RandomState(77)
    chk.random_sample(nd)
    assert np.random.random_sample() == chk.random_sample()
    W = r * cg.n_nodes
    early = float((ol < L).mean())
    assert early > 0.10, early
    print("directed sinks: %d walks, %.0f%% end early, %d passes" % (W, 100 * early, g2.stream_passes))
    assert g2.stream_passes < 0.35 * W, (g2.stream_passes, W)


@pytest.mark.parametrize("weighted,p,q", [(False, 0.25, 4.0), (True, 0.5, 2.0), (False, 0.3, 0.7)])
def test_budgeted_walk_lane_kernel_equals_table_walk(n2v, weighted, p, q):
    """Launches of >= ~2e5 walks of a partially stored engine run the lane-per-walk kernel (n2v_walk_otf.hip: stored steps
    per lane, rebuild steps served by the whole wave); smaller ones — every other budget test — the wave-per-walk one.
    240k walks on the 20k-node hub graph at a third of the tables, Philox (16-B bursts, L = 16) and numpy stream (L = 13:
    single-id writes), against the fully stored tables."""
    import torch
    rs = np.random.RandomState(12)
    n, m = 20000, 80000
    src = np.concatenate([rs.randint(0, n, size=m), np.full(700, 5), np.full(3000, 17)])
    dst = np.concatenate([rs.randint(0, n, size=m), rs.choice(n, 700, replace=False), rs.choice(n, 3000, replace=False)])
    keep = src != dst
    src, dst = src[keep], dst[keep]
    w = (rs.randint(1, 9, len(src)) / 2.0) if weighted else None
    from n2v_hip import csr
    cg = csr.from_edges(src, dst, w, False)
    g = n2v.Graph.from_csr(cg, p, q, rng="philox", seed=5)
    g.preprocess_transition_probs()
    full = g._engine.total_slots * 32
    a = g.simulate_walks(12, 16)
    g.rng = "numpy"
    np.random.seed(9)
    a2 = g.simulate_walks(12, 13)
    g.preprocess_transition_probs(budget_bytes=full // 3)
    assert g._engine.partial and int((~g._engine.stored_mask).sum()) > 0
    g.rng = "philox"
    b = g.simulate_walks(12, 16)
    assert torch.equal(a.lens, b.lens) and torch.equal(a.walks, b.walks)
    g.rng = "numpy"
    np.random.seed(9)
    b2 = g.simulate_walks(12, 13)
    assert torch.equal(a2.lens, b2.lens) and torch.equal(a2.walks, b2.walks)


@pytest.mark.parametrize("commons", [70, 126, 127, 128, 129, 300])
@pytest.mark.parametrize("p,q", [(0.25, 4.0), (0.3, 0.7)])
def test_on_the_fly_with_many_common_neighbours(n2v, commons, p, q):
    """Two hubs joined by an edge and sharing `commons` neighbours (plus private leaves): the class sweep of the
    on-the-fly walk keeps the positions of the weight-1 slots in a 128-entry list (prev's slot included) and hands rows
    with more of them to the table builder — both sides of that limit, dyadic and non-dyadic p, q, against the stored
    tables."""
    import torch
    from n2v_hip import csr
    src, dst = [0], [1]
    for c in range(2, 2 + commons):
        src += [0, 1]
        dst += [c, c]
    nxt = 2 + commons
    for hub, leaves in ((0, 90), (1, 150)):
        for _ in range(leaves):
            src.append(hub)
            dst.append(nxt)
            nxt += 1
    for c in range(2, 2 + commons, 3):                      # a few commons know each other: rows of <= 64 with adjacency
        src.append(c)
        dst.append(2 + (c - 2 + 1) % commons)
    cg = csr.from_edges(np.array(src), np.array(dst), None, False)
    g = n2v.Graph.from_csr(cg, p, q, rng="philox", seed=3)
    g.preprocess_transition_probs()
    a = g.simulate_walks(40, 12)
    g.force_on_the_fly = True
    b = g.simulate_walks_on_the_fly(40, 12)
    assert torch.equal(a.lens, b.lens) and torch.equal(a.walks, b.walks)


def test_randomised_parity_sweep(n2v):
    """40 random small graphs (directed or not, weighted or not, self-loops, isolated targets,
    duplicate lines, p and q from a grid incl. 1): tables and reference-exact walks vs the C oracle
    for every graph, vs the pure-Python oracle for every fourth; table-driven (thin and fat) and
    on-the-fly kernels."""
    import torch
    from n2v_hip import csr
    from oracle import c_oracle
    from oracle import n2v_oracle as orc
    rs = np.random.RandomState(77)
    grid = [0.25, 0.5, 1.0, 2.0, 4.0, 0.3]
    for trial in range(40):
        n = int(rs.randint(3, 60))
        m = int(rs.randint(2, 6 * n))
        directed, weighted = bool(rs.randint(2)), bool(rs.randint(2))
        src = rs.randint(0, n, m) * 3 + 1
        dst = rs.randint(0, n, m) * 3 + 1
        w = (rs.randint(1, 17, m) / 4.0) if weighted else None
        p, q = float(grid[rs.randint(6)]), float(grid[rs.randint(6)])
        cg = csr.from_edges(src, dst, w, directed)
        g = n2v.Graph.from_csr(cg, p, q, rng="numpy")
        g.preprocess_transition_probs()
        eng = g._engine
        co = c_oracle.CsrOracle(cg.row_ptr, cg.col, cg.w, p, q)
        co.preprocess(first_order_shortcut=eng.first_order)
        assert np.array_equal(eng.slots_J(eng.node_slots).cpu().numpy()[:cg.nnz], co.nodeJ), trial
        assert np.array_equal(_bits(eng.slots_q(eng.node_slots).cpu().numpy()[:cg.nnz]), _bits(co.nodeq)), trial
        if not eng.first_order:
            T = int(co.edge_off[-1])
            eJ, eq = eng.all_edge_tables()
            assert np.array_equal(eJ[:T], co.edgeJ), trial
            assert np.array_equal(_bits(eq[:T]), _bits(co.edgeq)), trial
        eng.preprocess(fat="both")
        if not eng.first_order:
            assert np.array_equal(eng.slots_J(eng.edge_slots).cpu().numpy()[:T], co.edgeJ), trial
            assert np.array_equal(_bits(eng.slots_q(eng.edge_slots).cpu().numpy()[:T]), _bits(co.edgeq)), trial
        r, L = int(rs.randint(1, 4)), int(rs.randint(1, 30))
        seed = int(rs.randint(1 << 30))
        ow, ol, nd = co.walk(cg.start_order, r, L, mode="mt", seed=seed)
        want = walks_from_padded(ow, ol, cg.labels)
        np.random.seed(seed)
        got = g.simulate_walks(r, L)
        assert got == want, (trial, directed, weighted, p, q)
        chk = np.random.RandomState(seed)
        chk.random_sample(nd)
        assert np.random.random_sample() == chk.random_sample(), trial
        g2 = n2v.Graph.from_csr(cg, p, q, rng="numpy")
        np.random.seed(seed)
        assert g2.simulate_walks_on_the_fly(r, L) == want, trial
        tw, tl = eng.walk(eng.start_order, r, max(L, 1), rng="philox", seed=seed, layout="thin")
        fw, fl = eng.walk(eng.start_order, r, max(L, 1), rng="philox", seed=seed, layout="fat")
        assert torch.equal(tw, fw) and torch.equal(tl, fl), trial
        if trial % 4 == 0:
            G = orc.OracleGraph(list(zip(src.tolist(), dst.tolist())), None if w is None else w.tolist(), directed)
            o = orc.Node2VecOracle(G, directed, p, q)
            assert o.simulate_walks(r, L, seed=seed, on_the_fly=True) == want, trial


def test_degrees_on_the_lds_window_boundaries(n2v):
    """Six graphs of tests/probes/boundary_stress.py (60 of them: profiles/r03/logs/boundary_stress_60_trials.log): hubs
    whose degrees sit on the LDS-window boundaries of the table builder (512 slots) and of the on-the-fly kernel (256),
    directed / weighted at random — every slot of the fat and thin tables, and the walks of the table kernel, the
    on-the-fly kernel and the budgeted-table kernel (Philox and numpy streams), against the C oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes"))
    import boundary_stress
    rs = np.random.RandomState(7)
    for t in range(6):
        boundary_stress.trial(t, rs)
