#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the reference itself.

Runs ONLY in the build container (where /root/reference is mounted read-only).
It imports the reference's own ``src/node2vec.py`` (with the one-line
``numpy.int = int`` shim the reference needs on numpy >= 1.24, applied in this
process, nothing is written into the reference tree) and dumps, per fixture
graph, everything SURVEY.md section 8(c) lists:

* ``nodes``            list(G.nodes())  (start order of simulate_walks)
* sorted adjacency     (the order alias slots refer to)
* ``alias_nodes``      every (J, q) as raw int64 / float64
* ``alias_edges``      every key + (J, q)
* walks                for several np.random seeds and (num_walks, walk_length)
                       shapes, for the ``nodes=`` subset call and for the
                       on-the-fly entry point, with the number of uniforms
                       the reference consumed.

The outputs are data (inputs + expected outputs); no reference source text is
stored.  Re-run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

np.int = int  # shim for /root/reference/src/node2vec.py:248 (np.int removed in numpy 1.24)
sys.dont_write_bytecode = True
REF_SRC = "/root/reference/src"
sys.path.insert(0, REF_SRC)
import networkx as nx  # noqa: E402
import node2vec as ref  # noqa: E402  (the reference module)

HERE = os.path.dirname(os.path.abspath(__file__))
KARATE = "/root/reference/graph/karate.edgelist"


def build_graph(edges, weights, directed):
    """Mirror src/main.py:66-80 (read_graph): DiGraph first, then to_undirected."""
    G = nx.DiGraph()
    for (u, v), w in zip(edges, weights):
        G.add_edge(int(u), int(v), weight=w)
    if not directed:
        G = G.to_undirected()
    return G


def count_draws(seed, state_after):
    """Number of doubles consumed since np.random.seed(seed)."""
    rs = np.random.RandomState(seed)
    key_after, pos_after = state_after[1], state_after[2]
    n = 0
    # each double = two 32-bit outputs; walk forward until the state matches
    while True:
        st = rs.get_state()
        if st[2] == pos_after and np.array_equal(st[1], key_after):
            return n
        rs.random_sample()
        n += 1
        if n > 50_000_000:
            raise RuntimeError("draw count not found")


def dump_case(name, edges, weights, directed, p, q, walk_specs, int_weights=False):
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    weights = np.asarray(weights, dtype=np.float64)
    wl = [int(w) for w in weights] if int_weights else [float(w) for w in weights]
    G = build_graph(edges, wl, directed)
    g = ref.Graph(G, directed, p, q)
    g.preprocess_transition_probs()

    nodes = list(G.nodes())
    out = {
        "edges": edges, "weights": weights, "directed": np.array(directed),
        "int_weights": np.array(int_weights),
        "p": np.array(float(p)), "q": np.array(float(q)),
        "nodes": np.array(nodes, dtype=np.int64),
    }
    adj_ptr = [0]
    adj, adj_w = [], []
    anJ, anq = [], []
    for v in nodes:
        nb = sorted(G.neighbors(v))
        adj.extend(nb)
        adj_w.extend(float(G[v][x]["weight"]) for x in nb)
        adj_ptr.append(len(adj))
        J, qq = g.alias_nodes[v]
        assert len(J) == len(nb)
        anJ.extend(int(x) for x in J)
        anq.extend(float(x) for x in qq)
    out["adj_ptr"] = np.array(adj_ptr, dtype=np.int64)
    out["adj"] = np.array(adj, dtype=np.int64)
    out["adj_w"] = np.array(adj_w, dtype=np.float64)
    out["an_J"] = np.array(anJ, dtype=np.int64)
    out["an_q"] = np.array(anq, dtype=np.float64)

    keys = list(g.alias_edges.keys())
    ae_ptr = [0]
    aeJ, aeq = [], []
    for k in keys:
        J, qq = g.alias_edges[k]
        aeJ.extend(int(x) for x in J)
        aeq.extend(float(x) for x in qq)
        ae_ptr.append(len(aeJ))
    out["ae_keys"] = np.array(keys, dtype=np.int64).reshape(-1, 2)
    out["ae_ptr"] = np.array(ae_ptr, dtype=np.int64)
    out["ae_J"] = np.array(aeJ, dtype=np.int64)
    out["ae_q"] = np.array(aeq, dtype=np.float64)

    metas = []
    for i, spec in enumerate(walk_specs):
        seed, r, L = spec["seed"], spec["r"], spec["L"]
        subset = spec.get("nodes")
        fly = spec.get("on_the_fly", False)
        np.random.seed(seed)
        fn = g.simulate_walks_on_the_fly if fly else g.simulate_walks
        walks = fn(r, L, nodes=subset)
        ndraws = count_draws(seed, np.random.get_state())
        flat = [x for w in walks for x in w]
        ptr = np.cumsum([0] + [len(w) for w in walks])
        assert ndraws == 2 * sum(len(w) - 1 for w in walks)
        out["walks_%d_flat" % i] = np.array(flat, dtype=np.int64)
        out["walks_%d_ptr" % i] = np.array(ptr, dtype=np.int64)
        out["walks_%d_subset" % i] = np.array(subset if subset else [], dtype=np.int64)
        metas.append([seed, r, L, ndraws, 1 if subset else 0, 1 if fly else 0])
    out["walk_meta"] = np.array(metas, dtype=np.int64).reshape(-1, 6)

    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s N=%d entries(nodes)=%d entries(edges)=%d walks=%d  %d B" % (
        name, len(nodes), len(anJ), len(aeJ), len(walk_specs), os.path.getsize(path)))


def karate_edges():
    with open(KARATE, "rb") as f:
        raw = f.read()
    assert hashlib.sha256(raw).hexdigest() == \
        "854f940aaf02b1755c916e31d31be2e0a50ef5a22a3f2b5bbe85236e5ff67808"
    e = [tuple(int(t) for t in line.split()[:2]) for line in raw.decode().splitlines() if line.strip()]
    return e


def std_specs(nodes_subset):
    return [
        {"seed": 123, "r": 10, "L": 80},
        {"seed": 7, "r": 10, "L": 80},
        {"seed": 123, "r": 2, "L": 10},
        {"seed": 123, "r": 1, "L": 1},
        {"seed": 5, "r": 1, "L": 2},
        {"seed": 11, "r": 3, "L": 20, "nodes": nodes_subset},
        {"seed": 123, "r": 2, "L": 30, "on_the_fly": True},
    ]


def alias_setup_vectors():
    """Known-answer vectors for alias_setup (src/node2vec.py:240-269)."""
    rs = np.random.RandomState(2024)
    probs_list = []
    # uniform tables incl. the K*(1/K) != 1 sizes (49, 98, 103, ...)
    for K in list(range(1, 131)) + [196, 206, 1000, 1999]:
        probs_list.append([1.0 / K] * K)
    # normalised random tables, the way the reference builds them
    for K in [1, 2, 3, 5, 8, 13, 34, 64, 65, 100, 257, 1024]:
        for _ in range(3):
            u = rs.random_sample(K) ** 3 + 1e-3
            norm = sum(float(x) for x in u)
            probs_list.append([float(x) / norm for x in u])
    # integer-like weights with three levels (1/p, 1, 1/q)
    for K in [4, 10, 33, 70, 300]:
        u = rs.choice([4.0, 1.0, 0.25], size=K)
        norm = sum(float(x) for x in u)
        probs_list.append([float(x) / norm for x in u])
    probs_list.append([])  # empty table
    ptr = [0]
    P, J, Q = [], [], []
    for pr in probs_list:
        j, qq = ref.alias_setup(pr)
        P.extend(pr)
        J.extend(int(x) for x in j)
        Q.extend(float(x) for x in qq)
        ptr.append(len(P))
    # alias_draw known answers: consume the global stream exactly as :277-278 do
    draws = []
    np.random.seed(99)
    j, qq = ref.alias_setup(probs_list[140])
    for _ in range(200):
        draws.append(int(ref.alias_draw(j, qq)))
    np.savez_compressed(
        os.path.join(HERE, "alias_setup.npz"),
        ptr=np.array(ptr, dtype=np.int64), probs=np.array(P, dtype=np.float64),
        J=np.array(J, dtype=np.int64), q=np.array(Q, dtype=np.float64),
        draw_table=np.array(140), draw_seed=np.array(99), draws=np.array(draws, dtype=np.int64))
    print("alias_setup vectors: %d tables, %d entries" % (len(probs_list), len(P)))


def main():
    alias_setup_vectors()

    ke = karate_edges()
    kw = [1] * len(ke)
    sub = [5, 1, 34, 12, 12, 3]
    dump_case("karate_p1_q1", ke, kw, False, 1, 1, std_specs(sub), int_weights=True)
    dump_case("karate_p025_q4", ke, kw, False, 0.25, 4, std_specs(sub), int_weights=True)
    dump_case("karate_p03_q07", ke, kw, False, 0.3, 0.7, std_specs(sub)[:3], int_weights=True)

    # weighted undirected toy, unordered labels, one label beyond int32, duplicate line
    we = [(10, 3), (3, 7), (7, 10), (7, 99999990001), (99999990001, 3), (5, 10), (5, 3),
          (2, 5), (2, 7), (10, 3), (8, 2), (8, 8), (8, 5)]
    ww = [0.5, 2.0, 1.5, 3.25, 0.125, 1.0, 7.0, 0.75, 2.5, 4.0, 1.0, 2.0, 0.3]
    dump_case("weighted_toy", we, ww, False, 0.5, 2.0, std_specs([8, 99999990001, 2]))

    # directed toy: sink (6), self-loop (4,4), asymmetric pairs, source-only node (0),
    # and a node only reachable as a target with no out-edges (9)
    de = [(0, 1), (1, 2), (2, 1), (2, 3), (3, 4), (4, 4), (4, 5), (5, 6), (1, 6), (3, 1),
          (5, 3), (2, 9), (7, 2), (7, 0), (4, 2)]
    dw = [1.0, 2.0, 1.0, 1.0, 3.0, 0.5, 1.0, 1.0, 0.25, 2.0, 1.0, 0.5, 1.0, 1.0, 4.0]
    dump_case("directed_toy", de, dw, True, 0.5, 2.0, std_specs([7, 0, 6, 4]))

    # star with 49 leaves: K*(1/K) = 0.9999999999999999 quirk on the hub's tables
    se = [(0, i) for i in range(1, 50)]
    dump_case("star49", se, [1] * 49, False, 1, 1, std_specs([0, 3, 49]), int_weights=True)

    # path graph with an isolated-by-construction tail: nodes of degree 1 (return-only tables)
    pe = [(i, i + 1) for i in range(12)]
    dump_case("path13", pe, [1] * 12, False, 2.0, 0.5, std_specs([0, 12, 6]), int_weights=True)

    # Erdos-Renyi G(n, m): 600 nodes, 3000 edges, seeded; not every id need appear
    rs = np.random.RandomState(42)
    n, m = 600, 3000
    seen = set()
    ee = []
    while len(ee) < m:
        u, v = int(rs.randint(n)), int(rs.randint(n))
        if u == v:
            continue
        key = (min(u, v), max(u, v))
        if key in seen:
            continue
        seen.add(key)
        ee.append((u, v))
    er_specs = [
        {"seed": 123, "r": 2, "L": 80},
        {"seed": 1, "r": 1, "L": 40},
        {"seed": 9, "r": 2, "L": 15, "nodes": [int(x) for x in rs.randint(n, size=50)]},
        {"seed": 123, "r": 1, "L": 25, "on_the_fly": True},
    ]
    dump_case("er600_p05_q2", ee, [1] * m, False, 0.5, 2.0, er_specs, int_weights=True)
    ew = (rs.random_sample(m) * 4 + 0.1).tolist()
    dump_case("er600_weighted_p4_q025", ee, ew, False, 4.0, 0.25, er_specs[:2])
    # directed ER with sinks: ragged walks, draw offsets depend on earlier lengths
    de2 = ee[:1200]
    dump_case("er600_directed", de2, (rs.random_sample(1200) + 0.5).tolist(), True, 0.25, 4.0,
              [{"seed": 123, "r": 2, "L": 30}, {"seed": 3, "r": 1, "L": 80}])

    # directed hub with 520 out-neighbours (> 512: the wave-per-table builder's in-place path and the on-the-fly
    # kernel's global scratch) reached over five in-edges, so only five tables are that large; cycles through the hub
    he, hw = [], []
    for t in range(1, 521):
        he.append((0, t)); hw.append(0.25 + (t * 37 % 11) / 4.0)
    for s_ in range(521, 526):
        he.append((s_, 0)); hw.append(1.0 + 0.5 * (s_ - 521))
    for t in range(1, 521):
        if t % 7 == 0:
            he.append((t, 521 + t % 5)); hw.append(0.75)
        if t % 13 == 0:
            he.append((t, 0)); hw.append(2.0)                      # straight back to the hub (p branch)
        if t % 5 == 0:
            he.append((t, (t * 3) % 520 + 1)); hw.append(1.5)      # target -> target: common-neighbour branch
    for s_ in range(521, 526):
        he.append((s_, (s_ * 17) % 520 + 1)); hw.append(0.6)
    dump_case("hub520_directed", he, hw, True, 0.25, 4.0,
              [{"seed": 123, "r": 2, "L": 40}, {"seed": 4, "r": 1, "L": 80, "nodes": [521, 0, 7, 522, 13, 525]},
               {"seed": 123, "r": 1, "L": 20, "on_the_fly": True}])


if __name__ == "__main__":
    main()
