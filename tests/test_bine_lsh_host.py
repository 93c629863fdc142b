"""CPU checks of oracle/bine_lsh_oracle.py — the restatement of datasketch 1.2.5 MinHash / MinHashLSHForest and of
src/bine_lsh.py:7-51 that the GPU pools are compared with.  Parity unpinned (datasketch is absent and the
reference holds no fixture for this path): what CAN be pinned independently is pinned here — SHA-1 against FIPS
180 vectors, the permutation parameters against numpy's legacy stream, the MinHash collision law, and the
array-based device algorithm (P) against the literal dictionary/sorted-list text (L)."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))

from oracle import bine_lsh_oracle as lo  # noqa: E402


def small_bipartite(n_u=60, n_v=40, seed=0, clones=12):
    """Users with overlapping item sets: a few prototypes, clones of them with one or two items changed, and
    exact duplicates — so that forest queries return more than the vertex itself."""
    rs = np.random.RandomState(seed)
    protos = [set(rs.choice(n_v, rs.randint(3, 9), replace=False).tolist()) for _ in range(n_u // 4)]
    rows = []
    for u in range(n_u):
        base = set(protos[rs.randint(len(protos))])
        if u % 3 == 1:
            base.add(int(rs.randint(n_v)))
        if u % 3 == 2 and len(base) > 2:
            base.discard(next(iter(base)))
        rows.append(sorted(base))
    users = np.concatenate([[u] * len(r) for u, r in enumerate(rows)])
    items = np.concatenate(rows)
    return users, items


def csr_of(users, items, n_u, n_v):
    from n2v_hip.bine import BipartiteGraph
    g = BipartiteGraph(["u%d" % u for u in users], ["i%d" % i for i in items], np.ones(len(users)))
    return g


def test_sha1_hash32_known_vectors():
    # FIPS 180-4 examples: SHA-1("abc") = a9993e36..., SHA-1("") = da39a3ee...
    assert lo.sha1_hash32(b"abc") == 0x363E99A9
    assert lo.sha1_hash32(b"") == 0xEEA339DA


def test_permutations_are_numpys_legacy_stream():
    a, b = lo.permutations()
    assert a.dtype == np.uint64 and a.shape == (128,)
    assert int(a.min()) >= 1 and int(a.max()) < lo.MERSENNE and int(b.max()) < lo.MERSENNE
    gen = np.random.RandomState(1)
    first = (gen.randint(1, lo.MERSENNE, dtype=np.uint64), gen.randint(0, lo.MERSENNE, dtype=np.uint64))
    assert (int(a[0]), int(b[0])) == (int(first[0]), int(first[1]))
    # the draws alternate a, b, a, b (one generator): a[1] is the THIRD draw
    third = gen.randint(1, lo.MERSENNE, dtype=np.uint64)
    assert int(a[1]) == int(third)


def test_minhash_collision_rate_is_jaccard():
    rs = np.random.RandomState(3)
    A = set(rs.choice(400, 120, replace=False).tolist())
    B = set(list(A)[:80]) | set(rs.choice(np.arange(400, 800), 40, replace=False).tolist())
    jac = len(A & B) / len(A | B)
    ma, mb = lo.MinHash(), lo.MinHash()
    for x in A:
        ma.update(str(x).encode())
    for x in B:
        mb.update(str(x).encode())
    rate = float(np.mean(ma.hashvalues == mb.hashvalues))
    assert abs(rate - jac) < 4.0 * np.sqrt(jac * (1 - jac) / 128)
    assert ma.hashvalues.max() <= lo.MAX_HASH


def test_device_algorithm_equals_literal_text():
    n_u, n_v = 60, 40
    users, items = small_bipartite(n_u, n_v)
    g = csr_of(users, items, n_u, n_v)
    labels = [str(x) for x in g.user_labels] + [str(x) for x in g.item_labels]
    hv = np.array([lo.sha1_hash32(s.encode("utf8")) for s in labels], dtype=np.uint64)
    for side_lo, side_hi in ((0, g.n_u), (g.n_u, g.n)):
        keys = labels[side_lo:side_hi]
        nbrs = [[labels[c] for c in g.col[g.row_ptr[v]:g.row_ptr[v + 1]]] for v in range(side_lo, side_hi)]
        for k in (5, 200):
            negs, info = lo.negs_by_lsh_literal(keys, nbrs, k=k, sample_num=7, rng=random.Random(1))
            sig = lo.signatures(g.row_ptr, g.col, hv, side_lo, side_hi)
            assert np.array_equal(sig.astype(np.uint64), np.stack(info["signatures"]))
            sims = lo.forest_query_all(sig, k=k)
            for i, s in info["sim"].items():
                assert {keys[j] for j in sims[i]} == s, (k, i)
            owner = lo.leaders(sims)
            for i, l in info["leader_of"].items():
                assert owner[i] == l
            for i, gone in info["excluded"].items():
                assert {keys[j] for j in lo.exclusions(sims, i)} == gone
            assert any(len(s) > 1 for s in sims)             # the graph does exercise the forest
            if k == 5:
                assert any(len(s) == 5 for s in sims)        # and the truncation at k
            for key, pool in negs.items():
                i = keys.index(key)
                assert not (set(pool) & info["excluded"][owner[i]])


def test_sample_pool_properties():
    gone = set(range(0, 300, 3))
    pool = lo.sample_pool(gone, 1000, 200, seed=5, leader=17)
    assert len(set(pool.tolist())) == 200 and not (set(pool.tolist()) & gone) and pool.min() >= 0
    assert not np.array_equal(pool, lo.sample_pool(gone, 1000, 200, seed=5, leader=18))
    short = lo.sample_pool(set(range(90)), 100, 20, seed=5, leader=0)
    assert short.tolist() == list(range(90, 100)) + [-1] * 10
