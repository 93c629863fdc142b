"""GPU parity tests of the MinHash LSH-forest negative pools (csrc/n2v_lsh.hip, include/n2v_bine.h) against
oracle/bine_lsh_oracle.py: label hashes == hashlib, signatures, every forest query (keys AND their order), the
clusters of the `visted` sweep and every pool row bit for bit.  The oracle restates datasketch 1.2.5 (absent here):
parity unpinned at the reference level, see the oracle header; tests/test_bine_lsh_host.py ties the array
algorithm checked here to the literal text of src/bine_lsh.py:7-51."""
import numpy as np
import pytest

from oracle import bine_lsh_oracle as lo

pytestmark = pytest.mark.gpu


def clustered_graph(n_u, n_v, seed, protos=None, label=str):
    """Users drawn from a few prototype item sets with small edits (plus exact duplicates): queries return
    neighbours, clusters form, some queries hit the truncation at k."""
    from n2v_hip import bine
    rs = np.random.RandomState(seed)
    protos = protos or max(n_u // 6, 2)
    sets = [set(rs.choice(n_v, rs.randint(3, 10), replace=False).tolist()) for _ in range(protos)]
    users, items = [], []
    for u in range(n_u):
        s = set(sets[rs.randint(protos)])
        if u % 3 == 1:
            s.add(int(rs.randint(n_v)))
        if u % 3 == 2 and len(s) > 2:
            s.discard(sorted(s)[0])
        for i in sorted(s):
            users.append(u)
            items.append(i)
    ul = [label("u%d" % u) for u in users]
    il = [label("i%d" % i) for i in items]
    return bine.BipartiteGraph(ul, il, np.ones(len(users)))


def engine(g, seed=11):
    from n2v_hip import bine
    return bine.BineEngine(g, device="cuda:0", seed=seed)


def labels_of(g):
    return [str(x) for x in g.user_labels] + [str(x) for x in g.item_labels]


def test_label_hashes_equal_hashlib():
    from n2v_hip import bine
    # short, 55/56/64-byte boundaries of the SHA-1 padding, two blocks, non-ASCII
    names = ["a", "abc", "x" * 55, "y" * 56, "z" * 63, "w" * 64, "v" * 119, "v" * 120, "k" * 200, "héllo", "节点7", ""]
    names = [n if n else "0" for n in names]
    g = bine.BipartiteGraph(names, ["i%d" % i for i in range(len(names))], np.ones(len(names)))
    e = engine(g)
    hv = e.label_hashes().cpu().numpy().view(np.uint32)
    want = np.array([lo.sha1_hash32(s.encode("utf8")) for s in labels_of(g)], dtype=np.uint32)
    assert np.array_equal(hv, want)


@pytest.mark.parametrize("n_u,n_v,k,pool_size,seed", [(150, 60, 200, 24, 1), (150, 60, 6, 10, 2), (40, 12, 200, 200, 3),
                                                      (700, 90, 20, 16, 4)])
def test_lsh_pools_equal_restatement(n_u, n_v, k, pool_size, seed):
    from n2v_hip import bine
    g = clustered_graph(n_u, n_v, seed)
    e = engine(g, seed=100 + seed)
    e.build_negative_pools(pool_size=pool_size, k=k)            # default method: the reference's LSH pipeline
    hv = np.array([lo.sha1_hash32(s.encode("utf8")) for s in labels_of(g)], dtype=np.uint64)
    sig_dev = e.lsh["signatures"].cpu().numpy().view(np.uint32)
    pool = e.pool.cpu().numpy()
    saw_cluster = saw_cut = saw_short = False
    for name, side_lo, side_hi, kseed in (("u", 0, g.n_u, bine.SEED_POOL_U), ("v", g.n_u, g.n, bine.SEED_POOL_V)):
        n_side = side_hi - side_lo
        sig = lo.signatures(g.row_ptr, g.col, hv, side_lo, side_hi)
        assert np.array_equal(sig_dev[side_lo:side_hi], sig)
        sims = lo.forest_query_all(sig, k=k)
        sim_dev = e.lsh[name]["sim"].cpu().numpy()
        cnt_dev = e.lsh[name]["sim_n"].cpu().numpy()
        for i, s in enumerate(sims):
            assert cnt_dev[i] == len(s) and sim_dev[i, :len(s)].tolist() == s, (name, i)
        owner = lo.leaders(sims)
        assert np.array_equal(e.lsh[name]["owner"].cpu().numpy(), owner)
        seed_side = bine.derive_seed(e.seed, kseed)
        rows = {}
        for i in range(n_side):
            l = int(owner[i])
            if l not in rows:
                gone = lo.exclusions(sims, l)
                rows[l] = lo.sample_pool(gone, n_side, pool_size, seed_side, l)
                saw_short = saw_short or rows[l].min() < 0
            want = np.where(rows[l] >= 0, rows[l] + side_lo, -1)
            assert np.array_equal(pool[side_lo + i], want), (name, i, l)
        saw_cluster = saw_cluster or bool(np.any(owner != np.arange(n_side)))
        saw_cut = saw_cut or any(len(s) == k for s in sims)
    assert saw_cluster
    if k < 50:
        assert saw_cut
    if n_u <= 40:
        assert saw_short


def test_integer_labels_and_training_with_short_pools():
    """Integer labels hash as their decimal text; a side smaller than the pool leaves -1 slots that the training
    pass skips (random.sample(negs, min(num_negs, len(negs))), src/bine_graph_utils.py:185)."""
    from n2v_hip import bine
    g0 = clustered_graph(60, 20, 5)
    users = np.array([int(str(x)[1:]) for x in g0.user_labels])[g0.edge_u]
    items = np.array([int(str(x)[1:]) for x in g0.item_labels])[g0.edge_v - g0.n_u] + 1000
    g = bine.BipartiteGraph(users, items, np.ones(len(users)))
    e = engine(g)
    hv = e.label_hashes().cpu().numpy().view(np.uint32)
    want = np.array([lo.sha1_hash32(str(int(x)).encode()) for x in list(g.user_labels) + list(g.item_labels)], dtype=np.uint32)
    assert np.array_equal(hv, want)
    e.calculate_centrality()
    e.generate_walks(percentage=0.15, maxT=4, minT=1)
    e.build_negative_pools(pool_size=64)
    assert int((e.pool < 0).sum()) > 0
    e.build_occurrences()
    e.init_embeddings(d=16)
    before = e.emb.clone()
    e.reset_schedule()
    e.train_pass(0, mode="sequential")
    assert bool(np.isfinite(e.emb.cpu().numpy()).all()) and not bool((e.emb == before).all())


def test_pools_never_hold_similar_vertices_on_a_larger_graph():
    """Property at a size the Python restatement does not reach: no pool entry is in the owner's exclusion set, rows
    are distinct vertices of the right side, clusters share rows."""
    g = clustered_graph(20000, 3000, 7, protos=2500)
    e = engine(g)
    e.build_negative_pools(pool_size=200)
    pool = e.pool.cpu().numpy()
    for name, side_lo, side_hi in (("u", 0, g.n_u), ("v", g.n_u, g.n)):
        sim = e.lsh[name]["sim"].cpu().numpy()
        sim_n = e.lsh[name]["sim_n"].cpu().numpy()
        owner = e.lsh[name]["owner"].cpu().numpy()
        rows = pool[side_lo:side_hi]
        assert rows.min() >= side_lo and rows.max() < side_hi
        assert np.array_equal(rows, rows[owner])
        assert np.all(owner[owner] == owner) and np.all(owner <= np.arange(side_hi - side_lo))
        srt = np.sort(rows, axis=1)
        assert np.all(srt[:, 1:] != srt[:, :-1])
        rs = np.random.RandomState(0)
        for l in rs.choice(np.unique(owner), 200, replace=False):
            gone = set(sim[l, :sim_n[l]].tolist())
            for j in sim[l, :sim_n[l]]:
                gone |= set(sim[j, :sim_n[j]].tolist())
            assert not (gone & set((rows[l] - side_lo).tolist()))
        assert int((sim_n > 1).sum()) > 0
