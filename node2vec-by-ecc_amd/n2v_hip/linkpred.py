"""Link-prediction scoring on the device (src/main_link.py:43-204 of the reference): batched cosine over
edge lists and ROC-AUC / average precision by rank statistics, so a 10^7-edge evaluation does not go through
a Python loop; and the top-k prediction path (``link_prediction`` / ``precision_at_k`` / ``calculate_pop``,
:62-170) on the tile + running-top-k kernels of csrc/n2v_sim.hip."""
import numpy as np
import torch

ITEM_PREFIX = "9999999"   # item nodes carry this id prefix (src/utils.py:392; tested at src/main_link.py:118-121,135)
KS = [1, 10, 50, 100, 500, 1000]   # default `ks` of link_prediction (src/main_link.py:123)


def link_score(emb, a, b, link_method="cos"):
    """src/main_link.py:43-60 on a KeyedVectors-like `emb`: "cos" -> float, "hadamard"/"avg" -> vectors (the
    reference returns them although its own roc_auc_score call cannot take them), "weight1" -> euclidean
    distance, "weight2" -> its square."""
    if link_method == "cos":
        try:
            return emb.similarity(str(a), str(b))
        except KeyError:
            print("something's wrong. a:{}, b:{}".format(a, b))
            return 0
    x, y = emb[str(a)], emb[str(b)]
    if link_method == "hadamard":
        return np.multiply(x, y)
    if link_method == "avg":
        return (x + y) / 2.0
    diff = x - y
    if link_method == "weight1":
        return np.sqrt(diff.dot(diff))
    if link_method == "weight2":
        return diff.dot(diff)
    raise ValueError("link_method %r" % (link_method,))


def cosine_scores(vectors, pairs):
    """vectors: float32 [N, d] device tensor (row = dense id); pairs: int64 [M, 2] dense ids."""
    a = vectors[pairs[:, 0]]
    b = vectors[pairs[:, 1]]
    na = a.norm(dim=1).clamp_min(1e-30)
    nb = b.norm(dim=1).clamp_min(1e-30)
    return (a * b).sum(dim=1) / (na * nb)


def roc_auc(pos_scores, neg_scores):
    """Mann-Whitney U with midranks == sklearn.metrics.roc_auc_score."""
    s = torch.cat([pos_scores, neg_scores]).double()
    n_pos, n_neg = pos_scores.numel(), neg_scores.numel()
    vals, inv, counts = torch.unique(s, sorted=True, return_inverse=True, return_counts=True)
    csum = torch.cumsum(counts, 0).double()
    midrank = csum - (counts.double() - 1) / 2.0  # 1-based average rank of each distinct value
    ranks = midrank[inv]
    u = ranks[:n_pos].sum() - n_pos * (n_pos + 1) / 2.0
    return float(u / (n_pos * n_neg))


def average_precision(pos_scores, neg_scores):
    """sklearn.metrics.average_precision_score (step-wise, ties grouped by threshold)."""
    s = torch.cat([pos_scores, neg_scores]).double()
    y = torch.cat([torch.ones_like(pos_scores), torch.zeros_like(neg_scores)]).double()
    order = torch.argsort(s, descending=True, stable=True)
    s, y = s[order], y[order]
    tp = torch.cumsum(y, 0)
    last = torch.ones_like(s, dtype=torch.bool)
    last[:-1] = s[1:] != s[:-1]  # last element of each tie group = one threshold
    tp_t = tp[last]
    k_t = (torch.nonzero(last).squeeze(1) + 1).double()
    precision = tp_t / k_t
    recall = tp_t / y.sum()
    prev = torch.cat([torch.zeros(1, dtype=torch.double, device=s.device), recall[:-1]])
    return float(((recall - prev) * precision).sum())


def get_roc_score(vectors, edges_pos, edges_neg):
    """(roc_auc, average_precision) of cosine scores; edges are dense-id pairs."""
    d = vectors.device
    pos = cosine_scores(vectors, torch.as_tensor(np.asarray(edges_pos), dtype=torch.int64, device=d))
    neg = cosine_scores(vectors, torch.as_tensor(np.asarray(edges_neg), dtype=torch.int64, device=d))
    return roc_auc(pos, neg), average_precision(pos, neg)


# --------------------------------------------------------------------------- the main_link flow
def split_edges(edges, test_ratio=0.5, seed=123):
    """src/main_link.py:525-526 with src/settings.py:1-2: sklearn's train_test_split on the
    (u, v) edge array (TEST_RATIO 0.5, RANDOM_SEED 123)."""
    from sklearn.model_selection import train_test_split
    tr, te = train_test_split(np.asarray(edges), test_size=test_ratio, random_state=seed)
    return tr, te


def build_neg_samples(labels, true_edges, seed=0):
    """src/main_link.py:191-204: as many distinct non-edges (a < b) as there are edges, drawn
    uniformly over node pairs — vectorised rejection sampling so 10^7 edges do not go through
    a Python loop; the reference's `random.sample` is unseeded, a seed is taken here."""
    labels = np.asarray(labels, dtype=np.int64)
    e = np.asarray(true_edges, dtype=np.int64).reshape(-1, 2)
    n = len(labels)
    ia, ib = np.searchsorted(labels, e[:, 0]), np.searchsorted(labels, e[:, 1])
    true_key = np.unique(np.minimum(ia, ib) * n + np.maximum(ia, ib))
    want = len(true_key)
    rs = np.random.RandomState(seed)
    got = np.zeros(0, dtype=np.int64)
    while len(got) < want:
        k = int((want - len(got)) * 1.3) + 64
        a, b = rs.randint(0, n, size=k), rs.randint(0, n, size=k)
        key = np.minimum(a, b) * np.int64(n) + np.maximum(a, b)
        key = key[a != b]
        key = key[~np.isin(key, true_key, assume_unique=False)]
        got = np.unique(np.concatenate([got, key]))
        if len(got) > want:
            got = rs.permutation(got)[:want]
    return np.stack([labels[got // n], labels[got % n]], 1)


def run(edges, p=1.0, q=1.0, num_walks=5, walk_length=40, dimensions=128, window_size=10, iter=1,
        directed=False, test_ratio=0.5, split_seed=123, neg_seed=0, rng="philox", seed=1, device=None,
        add_user_edges=False, user_edges_mode="ratio", user_edges_ratio=0.1, user_edges_thre=0.5,
        unseparated=False):
    """The AUC path of src/main_link.py:519-563 (defaults of src/settings.py: 5 walks of length
    40, d=128): split the edges 50/50, walk and embed on the TRAINING graph only (nodes isolated
    by the removal keep their length-1 walks), score test edges against sampled non-edges of
    the full graph by cosine similarity.  Returns {'roc': ..., 'ap': ..., ...}."""
    import node2vec
    from . import csr, sgns
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    full = csr.from_edges(edges[:, 0], edges[:, 1], None, directed)
    tr, te = split_edges(edges, test_ratio, split_seed)
    # nx_G.remove_edges_from(test_edges): the node set (and its order) stays the full graph's
    train = csr.from_edges(tr[:, 0], tr[:, 1], None, directed)
    missing = np.setdiff1d(full.labels, train.labels)
    if len(missing):
        train = _with_isolated_nodes(train, full)
    g = node2vec.Graph.from_csr(train, p, q, device=device, rng=rng, seed=seed)
    g.preprocess_transition_probs()
    corpus = g.simulate_walks(num_walks, walk_length)
    model = sgns.SgnsModel(train.n_nodes, dim=dimensions, window=window_size, seed=seed, device=corpus.walks.device)
    model.build_vocab(corpus.walks)
    sgns.train(model, corpus.walks, corpus.lens, epochs=iter)
    neg = build_neg_samples(full.labels, edges, neg_seed)
    te_d = np.stack([train.dense_of(te[:, 0]), train.dense_of(te[:, 1])], 1)
    neg_d = np.stack([train.dense_of(neg[:, 0]), train.dense_of(neg[:, 1])], 1)
    roc, ap = get_roc_score(model.vectors(), te_d, neg_d)
    out = {"roc": roc, "ap": ap, "n_nodes": int(full.n_nodes), "n_train": int(len(tr)), "n_test": int(len(te)),
           "pairs_trained": model.pairs_trained(), "model": model, "graph": g, "roc_user": None, "ap_user": None}
    if add_user_edges:
        # src/main_link.py:568-599: similarity edges between user nodes, then walk, embed and score again
        from . import augment
        train2, n_added = augment.augment_graph(train, model.vectors(), user_edges_mode, user_edges_ratio,
                                                user_edges_thre, unseparated)
        g2 = node2vec.Graph.from_csr(train2, p, q, device=device, rng=rng, seed=seed)
        g2.preprocess_transition_probs()
        corpus2 = g2.simulate_walks(num_walks, walk_length)
        model2 = sgns.SgnsModel(train2.n_nodes, dim=dimensions, window=window_size, seed=seed, device=corpus2.walks.device)
        model2.build_vocab(corpus2.walks)
        sgns.train(model2, corpus2.walks, corpus2.lens, epochs=iter)
        out["roc_user"], out["ap_user"] = get_roc_score(model2.vectors(), te_d, neg_d)
        out.update({"edges_added": n_added, "model_user": model2, "graph_user": g2})
    return out


def _with_isolated_nodes(train, full):
    """Re-express the training CSR over the full graph's node set (same labels / start order),
    so nodes that lost all their edges stay in the graph with degree 0."""
    from .csr import CsrGraph
    pos = np.searchsorted(full.labels, train.labels)
    deg = np.zeros(full.n_nodes, dtype=np.int64)
    deg[pos] = np.diff(train.row_ptr)
    row_ptr = np.zeros(full.n_nodes + 1, dtype=np.int64)
    np.cumsum(deg, out=row_ptr[1:])
    col = pos[train.col].astype(np.int32)   # rows stay sorted: pos is increasing
    return CsrGraph(full.labels, row_ptr, col, train.w, full.start_order, train.directed)


# --------------------------------------------------------------------------- top-k prediction (:62-170)
def _is_item(labels):
    return np.array([str(int(x)).startswith(ITEM_PREFIX) for x in labels], dtype=bool)


def precision_at_k(pred_k, test_edges):
    """src/main_link.py:62-67: share of the predicted pairs that are test edges in either orientation."""
    test = set((a, b) for a, b in test_edges)
    count = 0.0
    for pred in pred_k:
        if tuple(pred) in test or (pred[1], pred[0]) in test:
            count += 1
    return count / len(pred_k)


def calculate_pop(degrees_by_label, chosen_k_links, unseparated=False):
    """src/main_link.py:110-120: "popularity" = len(g[node]) of the item end of a link (mean of both ends,
    floored, when unseparated).  degrees_by_label: callable label -> degree in the training graph."""
    pop = []
    for a, b in chosen_k_links:
        a, b = int(a), int(b)
        if unseparated:
            pop.append(int((degrees_by_label(a) + degrees_by_label(b)) // 2))
        elif str(a).startswith(ITEM_PREFIX):
            pop.append(degrees_by_label(a))
        elif str(b).startswith(ITEM_PREFIX):
            pop.append(degrees_by_label(b))
    assert len(pop) == len(chosen_k_links)
    return pop


def link_prediction(vectors, graph, train_edges, test_edges, ks=KS, unseparated=False, vocab_mask=None,
                    sim_method="cos"):
    """src/main_link.py:123-170.  vectors: device float [N, >=d] indexed by dense id of `graph` (the training
    CsrGraph: dense ids ascend with the label, i.e. the reference's `sorted(int(x) for x in emb.vocab)` order);
    train_edges / test_edges: int arrays [M, 2] of labels.  Every (user, item) pair — or every pair
    nodes[i], nodes[j], i < j, when `unseparated` — that is not a training edge IN THAT ORIENTATION (the
    reference subtracts the tuples as they are, :74,:84) is scored on the device; the max(ks) best survive a
    running threshold, so nothing of size users x items is stored.  The reference's `segment` batching only
    bounds its memory: merging per-segment top-k lists and cutting at k again is the global top-k.
    Returns (results, final_results) as the reference: results[k] = [((a, b), score, pop), ...] with string
    ids, final_results[k] = (precision, avg_pop) — avg_pop floored like the reference's Python-2 division."""
    from . import simsel
    labels = graph.labels
    n = len(labels)
    in_vocab = np.ones(n, dtype=bool) if vocab_mask is None else np.asarray(vocab_mask, dtype=bool)
    if unseparated:
        rows = cols = np.nonzero(in_vocab)[0]
    else:
        item = _is_item(labels)
        rows, cols = np.nonzero(in_vocab & ~item)[0], np.nonzero(in_vocab & item)[0]
    dev = vectors.device
    dim = int(vectors.shape[1])
    A = simsel.prepare(vectors, sim_method, rows=torch.from_numpy(rows), dim=dim)
    B = A if unseparated else simsel.prepare(vectors, sim_method, rows=torch.from_numpy(cols), dim=dim)
    n_rows, n_cols = len(rows), len(cols)
    # training edges, as ordered (row, col) keys
    tr = np.asarray(train_edges, dtype=np.int64).reshape(-1, 2)
    pos_r, pos_c = np.full(n, -1, dtype=np.int64), np.full(n, -1, dtype=np.int64)
    pos_r[rows] = np.arange(n_rows)
    pos_c[cols] = np.arange(n_cols)
    known = np.isin(tr, labels).all(axis=1) if len(tr) else np.zeros(0, dtype=bool)
    da, db = graph.dense_of(tr[known, 0]), graph.dense_of(tr[known, 1])
    ok = (pos_r[da] >= 0) & (pos_c[db] >= 0)
    if unseparated:
        ok &= pos_r[da] < pos_c[db]          # only pairs nodes[i], nodes[j] with i < j are candidates (:72)
    keys = np.unique(pos_r[da[ok]] * n_cols + pos_c[db[ok]])
    total = n_rows * (n_rows - 1) // 2 if unseparated else n_rows * n_cols
    kmax = int(min(max(ks), max(total - len(keys), 0)))
    if kmax == 0:
        raise ValueError("link_prediction: no candidate pairs")
    s, r, c = simsel.global_topk(A, B, kmax, sim_method, upper_triangle=unseparated,
                                 exclude_keys=torch.from_numpy(keys).to(dev))
    s, r, c = s.cpu().numpy(), rows[r.cpu().numpy()], cols[c.cpu().numpy()]
    deg = np.diff(graph.row_ptr)
    te = np.asarray(test_edges, dtype=np.int64).reshape(-1, 2)
    te_known = np.isin(te, labels).all(axis=1) if len(te) else np.zeros(0, dtype=bool)
    ta, tb = graph.dense_of(te[te_known, 0]), graph.dense_of(te[te_known, 1])
    test_keys = np.unique(np.minimum(ta, tb) * n + np.maximum(ta, tb))
    hit = np.isin(np.minimum(r, c).astype(np.int64) * n + np.maximum(r, c), test_keys)
    item_mask = None if unseparated else _is_item(labels)
    if unseparated:
        pop = (deg[r] + deg[c]) // 2
    else:
        pop = np.where(item_mask[r], deg[r], deg[c])
    results, final_results = {}, {}
    for k in ks:
        kk = min(k, len(s))
        results[k] = [((str(int(labels[r[i]])), str(int(labels[c[i]]))), float(s[i]), int(pop[i])) for i in range(kk)]
        final_results[k] = (float(hit[:kk].sum()) / kk, int(pop[:kk].sum()) // kk)
    return results, final_results
