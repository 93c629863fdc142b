"""Stress of the round-3 table builder / on-the-fly / budgeted-table code on graphs whose degrees sit on the LDS-window
boundaries (255 ... 257 slots of the on-the-fly kernel, 511 ... 513 of the builder, hubs of a few thousand): every slot of
every table and every walk against the C oracle.  python tests/probes/boundary_stress.py [trials]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import node2vec as n2v
from n2v_hip import csr
from oracle import c_oracle


def bits(x):
    return np.ascontiguousarray(x, dtype=np.float64).view(np.uint64)


def trial(t, rs):
    n = int(rs.randint(3100, 7000))
    m = int(rs.randint(2 * n, 6 * n))
    directed, weighted = bool(rs.randint(2)), bool(rs.randint(2))
    src, dst = [rs.randint(0, n, m)], [rs.randint(0, n, m)]
    for deg in rs.choice([255, 256, 257, 300, 511, 512, 513, 700, 1200, 3000], size=int(rs.randint(2, 6)), replace=False):
        hub = int(rs.randint(0, n))
        nb = rs.choice(n, int(deg), replace=False)
        src.append(np.full(deg, hub)); dst.append(nb)
        if directed and rs.randint(2):
            src.append(nb[: deg // 3]); dst.append(np.full(deg // 3, hub))
    src, dst = np.concatenate(src), np.concatenate(dst)
    if rs.randint(2):
        keep = src != dst
        src, dst = src[keep], dst[keep]
    w = (rs.randint(1, 17, len(src)) / 4.0) if weighted else None
    grid = [0.25, 0.5, 2.0, 4.0, 0.3, 1.0]
    p, q = float(grid[rs.randint(6)]), float(grid[rs.randint(5)])
    cg = csr.from_edges(src, dst, w, directed)
    g = n2v.Graph.from_csr(cg, p, q, rng="philox", seed=1000 + t)
    g.preprocess_transition_probs()
    eng = g._engine
    co = c_oracle.CsrOracle(cg.row_ptr, cg.col, cg.w, p, q)
    co.preprocess(first_order_shortcut=eng.first_order)
    assert np.array_equal(eng.slots_J(eng.node_slots).cpu().numpy()[:cg.nnz], co.nodeJ)
    if not eng.first_order:
        T = int(co.edge_off[-1])
        eJ, eq = eng.all_edge_tables()
        assert np.array_equal(eJ[:T], co.edgeJ), "J"
        assert np.array_equal(bits(eq[:T]), bits(co.edgeq)), "q"
        eng.preprocess(fat=False)                                    # thin output of the wave builder
        assert np.array_equal(eng.slots_J(eng.edge_slots).cpu().numpy()[:T], co.edgeJ)
        assert np.array_equal(bits(eng.slots_q(eng.edge_slots).cpu().numpy()[:T]), bits(co.edgeq))
        eng.preprocess()
    r, L = int(rs.randint(1, 3)), int(rs.choice([16, 23, 32, 40]))
    ow, ol, _ = co.walk(cg.start_order, r, L, mode="philox", seed=1000 + t)
    c = g.simulate_walks(r, L)
    assert np.array_equal(c.walks.cpu().numpy(), ow) and np.array_equal(c.lens.cpu().numpy(), ol), "fat walk"
    fw, fl = eng.walk_on_the_fly(eng.start_order, r, L, rng="philox", seed=1000 + t)
    assert np.array_equal(fw.cpu().numpy(), ow) and np.array_equal(fl.cpu().numpy(), ol), "on the fly"
    if not eng.first_order:
        full = eng.total_slots * 32
        g.preprocess_transition_probs(budget_bytes=int(full * rs.uniform(0.02, 0.9)))
        c = g.simulate_walks(r, L)
        assert np.array_equal(c.walks.cpu().numpy(), ow), "hybrid"
        g.rng = "numpy"
        seed = int(rs.randint(1 << 30))
        np.random.seed(seed)
        c = g.simulate_walks(1, L)
        mw, ml, _ = co.walk(cg.start_order, 1, L, mode="mt", seed=seed)
        assert np.array_equal(c.walks.cpu().numpy(), mw), "hybrid, numpy stream"
    return n, len(src), directed, weighted, p, q, int(cg.degrees.max())


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rs = np.random.RandomState(2026)
    for t in range(trials):
        info = trial(t, rs)
        print("trial %d ok: n %d, m %d, directed %s, weighted %s, p %.2f q %.2f, max degree %d" % ((t,) + info), flush=True)
