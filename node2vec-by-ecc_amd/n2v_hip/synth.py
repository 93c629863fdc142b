"""Seeded synthetic graphs for the benchmark configs of BASELINE.md section 4 (C2: Erdos-Renyi
G(n, m); C3: Barabasi-Albert preferential attachment).  Pure numpy, deterministic for a
seed, fast enough to regenerate on the GPU box (no dataset travels)."""
import hashlib

import numpy as np

from . import csr


def erdos_renyi_edges(n, m, seed=42):
    """m distinct unordered pairs without self-loops, ids 0..n-1."""
    rs = np.random.RandomState(seed)
    got_u = np.zeros(0, dtype=np.int64)
    got_v = np.zeros(0, dtype=np.int64)
    while len(got_u) < m:
        k = int((m - len(got_u)) * 1.2) + 16
        u = rs.randint(0, n, size=k).astype(np.int64)
        v = rs.randint(0, n, size=k).astype(np.int64)
        u = np.concatenate([got_u, u])
        v = np.concatenate([got_v, v])
        ok = u != v
        u, v = u[ok], v[ok]
        key = np.minimum(u, v) * n + np.maximum(u, v)
        _, first = np.unique(key, return_index=True)
        first.sort()
        got_u, got_v = u[first], v[first]
    return got_u[:m], got_v[:m]


def barabasi_albert_edges(n, m, seed=42):
    """Preferential attachment by the Batagelj-Brandes copy model: edge i has source
    i // m + m0 ... and its target copies a uniformly random earlier endpoint, which is
    attachment proportional to degree.  Pointer chains through not-yet-resolved targets
    are resolved by pointer jumping (vectorised).  Self-loops and parallel edges are
    dropped, so |E| is slightly below n*m (~1e7 edges for n=1e6, m=10)."""
    rs = np.random.RandomState(seed)
    E = (n - 1) * m
    idx = np.arange(E, dtype=np.int64)
    src = idx // m + 1                      # node 0 is the seed node
    # endpoint slots: 2i = source of edge i, 2i+1 = target of edge i; node 0 gets a virtual
    # slot so that the first edges have something to attach to
    r = np.floor(rs.random_sample(E) * (2 * idx + 1)).astype(np.int64) - 1   # -1 = seed node
    ptr = r.copy()
    for _ in range(200):
        odd = (ptr >= 0) & (ptr & 1 == 1)
        if not odd.any():
            break
        ptr[odd] = r[ptr[odd] >> 1]
    else:
        raise RuntimeError("pointer jumping did not converge")
    dst = np.where(ptr < 0, 0, src[np.maximum(ptr, 0) >> 1])
    ok = src != dst
    src, dst = src[ok], dst[ok]
    key = np.minimum(src, dst) * n + np.maximum(src, dst)
    _, first = np.unique(key, return_index=True)
    first.sort()
    return src[first], dst[first]


def edges_sha256(u, v):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(u, dtype=np.int64).tobytes())
    h.update(np.ascontiguousarray(v, dtype=np.int64).tobytes())
    return h.hexdigest()


def make_config_graph(name):
    """'C2' -> ER(100k, 1M); 'C3' -> BA(1M, m=10); returns (CsrGraph, info dict)."""
    if name == "C2":
        u, v = erdos_renyi_edges(100_000, 1_000_000, 42)
        kind = "erdos_renyi n=100000 m=1000000 seed=42"
    elif name == "C3":
        u, v = barabasi_albert_edges(1_000_000, 10, 42)
        kind = "barabasi_albert n=1000000 m=10 seed=42"
    elif name.startswith("ba:"):
        _, n, m = name.split(":")
        u, v = barabasi_albert_edges(int(n), int(m), 42)
        kind = "barabasi_albert n=%s m=%s seed=42" % (n, m)
    elif name.startswith("er:"):
        _, n, m = name.split(":")
        u, v = erdos_renyi_edges(int(n), int(m), 42)
        kind = "erdos_renyi n=%s m=%s seed=42" % (n, m)
    else:
        raise ValueError(name)
    g = csr.from_edges(u, v, None, directed=False)
    deg = g.degrees
    info = {"graph": kind, "nodes": int(g.n_nodes), "edges": int(len(u)), "nnz": int(g.nnz),
            "sum_deg2": int((deg.astype(np.int64) ** 2).sum()), "max_deg": int(deg.max()),
            "edges_sha256": edges_sha256(u, v)}
    return g, info


def bipartite_powerlaw_ratings(n_users, n_items, n_ratings, exponent=0.75, seed=42):
    """BASELINE config 5's shape: every user rates n_ratings // n_users items drawn from a power-law item
    popularity p(rank) ~ (rank + 1)^-exponent (duplicate pairs kept: the rating list may repeat a pair, as a
    ratings file may), ratings uniform in {1..5}.  Returns (users, items, ratings) in user-major order; items
    that nobody rated do not appear (n_items is the size of the catalogue drawn from)."""
    rs = np.random.RandomState(seed)
    per_user = max(1, n_ratings // n_users)
    cdf = np.cumsum(1.0 / np.power(np.arange(1, n_items + 1, dtype=np.float64), exponent))
    cdf /= cdf[-1]
    users = np.repeat(np.arange(n_users, dtype=np.int64), per_user)
    items = np.searchsorted(cdf, rs.random_sample(users.shape[0])).astype(np.int64)
    np.minimum(items, n_items - 1, out=items)
    ratings = rs.randint(1, 6, size=users.shape[0]).astype(np.float64)
    return users, items, ratings
