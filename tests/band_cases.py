"""Graphs and link-prediction splits of the SGNS acceptance-band tests (BASELINE.json: AUC within +-0.002 of the
sequential comparator), shared by the fixture generator tests/golden/make_sgns_band.py (CPU only, build container)
and the -m gpu tests that train on the GPU-generated — bit-identical, checked by hash — walks.

Everything here is host-side numpy: the generators are seeded, so the GPU box rebuilds the same graph, the same
50/50 split (src/main_link.py:525-526, seed 123) and the same negative pairs (src/main_link.py:191-204)."""
import hashlib
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BAND_DIR = os.path.join(ROOT, "tests", "golden", "sgns_band")


def planted_partition(n=3000, k=30, m_in=20000, m_out=4000, seed=0):
    """Uniform-degree planted partition (k communities)."""
    rs = np.random.RandomState(seed)
    comm = rs.randint(0, k, n)
    src, dst = [], []
    while len(src) < m_in:
        a, b = rs.randint(0, n, 2)
        if a != b and comm[a] == comm[b]:
            src.append(a)
            dst.append(b)
    for _ in range(m_out):
        a, b = rs.randint(0, n, 2)
        if a != b:
            src.append(a)
            dst.append(b)
    src, dst = np.array(src), np.array(dst)
    key = np.minimum(src, dst) * n + np.maximum(src, dst)
    _, first = np.unique(key, return_index=True)
    first.sort()
    return np.stack([src[first], dst[first]], 1)


def hub_partition(n=20000, k=100, m_in=200000, m_out=40000, seed=0):
    """Degree-corrected planted partition: communities + Pareto node activity (hubs) — the hub-heavy counterpart
    (C3/C4 are power-law graphs)."""
    rs = np.random.RandomState(seed)
    comm = rs.randint(0, k, n)
    theta = rs.pareto(1.5, n) + 1.0
    order = np.argsort(comm, kind="stable")
    starts = np.searchsorted(comm[order], np.arange(k + 1))
    src, dst = [], []
    per = m_in // k
    for c in range(k):
        members = order[starts[c]:starts[c + 1]]
        if len(members) < 2:
            continue
        pr = theta[members] / theta[members].sum()
        src.append(rs.choice(members, per, p=pr))
        dst.append(rs.choice(members, per, p=pr))
    pr = theta / theta.sum()
    src.append(rs.choice(n, m_out, p=pr))
    dst.append(rs.choice(n, m_out, p=pr))
    src, dst = np.concatenate(src), np.concatenate(dst)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.minimum(src, dst) * n + np.maximum(src, dst)
    _, first = np.unique(key, return_index=True)
    first.sort()
    return np.stack([src[first], dst[first]], 1)


# name -> (graph generator, its arguments, rounds, walk length).  Walk seed 1 (Philox), d = 128, window 10,
# negative 5, SGNS seed 1 — the arguments of src/main.py:82-90 with gensim's defaults.
CASES = {
    "uniform3k_10x80": ("planted", {}, 10, 80),
    "hub20k_10x80": ("hub", {}, 10, 80),
    "hub131k_10x80": ("hub", dict(n=131072, k=131072 // 200, m_in=10 * 131072, m_out=2 * 131072, seed=2), 10, 80),
    "hub131k_5x40": ("hub", dict(n=131072, k=131072 // 200, m_in=10 * 131072, m_out=2 * 131072, seed=2), 5, 40),
    # above the `auto` switch to agent-scope rows (>= 131 072 rows, >= 600 tokens per row); comparator: ~4 h of one core
    "hub400k_10x80": ("hub", dict(n=400000, k=2000, m_in=4000000, m_out=800000, seed=3), 10, 80),
}


# graphs of probes only (no comparator fixture: the sequential comparator would need ~10 h of one core)
PROBE_CASES = {
    "hub1m_10x80": ("hub", dict(n=1000000, k=5000, m_in=10000000, m_out=2000000, seed=4), 10, 80),   # C3's size
}


def sha16(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()[:16]


def build(name):
    """-> dict(graph=CsrGraph of the TRAINING edges over the full node set, te_d, neg_d = dense-id test / negative
    pairs, rounds, L, edges_sha).  src/main_link.py:519-563: the test edges are removed, the node set stays."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
    from n2v_hip import csr, linkpred
    kind, kw, rounds, L = (CASES.get(name) or PROBE_CASES[name])
    edges = (planted_partition if kind == "planted" else hub_partition)(**kw)
    tr, te = linkpred.split_edges(edges)
    full = csr.from_edges(edges[:, 0], edges[:, 1], None, False)
    g = csr.from_edges(tr[:, 0], tr[:, 1], None, False)
    if g.n_nodes != full.n_nodes:
        g = linkpred._with_isolated_nodes(g, full)
    neg = linkpred.build_neg_samples(full.labels, edges, 0)
    te_d = np.stack([g.dense_of(te[:, 0]), g.dense_of(te[:, 1])], 1)
    neg_d = np.stack([g.dense_of(neg[:, 0]), g.dense_of(neg[:, 1])], 1)
    return dict(graph=g, te_d=te_d, neg_d=neg_d, rounds=rounds, L=L, edges_sha=sha16(edges.astype(np.int64)))


def fixture_path(name):
    return os.path.join(BAND_DIR, name + ".json")


def load_fixture(name):
    with open(fixture_path(name)) as f:
        return json.load(f)
