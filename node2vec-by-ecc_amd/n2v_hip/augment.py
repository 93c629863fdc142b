"""Similarity-driven edge augmentation (src/main_link.py:358-475, 568-599):
after a first embedding, connect every "user" node to the users most similar to it, then walk
and embed again on the augmented, now weighted, graph.

The reference fills an N_user x N_user similarity matrix with a Python double loop over
``get_similarity`` (:358-377: "cos" = gensim similarity, "pearson" = scipy pearsonr, "jsd" = its js()) and
sorts every row in Python (:379-453).  Here the tile kernel of csrc/n2v_sim.hip forms the scores of a block
of users against all users once and one workgroup per user selects on the device — by threshold, or by an
exact radix select of the int(N*ratio)-th largest score — so nothing of size N^2 is ever held.
Semantics kept from the reference: a user's similarity to itself counts as 0
(:386,:404) and is NOT excluded from the ranking; "ratio" keeps the int(N*ratio) most similar per
user in descending order (ties in list order) with weight 1; "step" keeps similarity > threshold
with weight 1; "relu" the same with weight = similarity; "relu-ratio" is "relu" with the ratio as
its threshold (that is what :469 does); "linear" keeps every pair with weight = similarity.
"jsd" is kept as the reference has it: a DIVERGENCE used as if it were a similarity, infinite whenever a
vector has a negative component (scipy rel_entr) — i.e. for practically every learned embedding.
PARITY UNPINNED: the reference's main_link.py does not import here (pathos, gensim), its output
is pinned by no fixture; the tests compare against a plain-Python restatement of that text.
"""
import numpy as np
import torch

from . import simsel

ITEM_PREFIX = "9999999"  # item nodes are marked by this id prefix (src/utils.py:392, main_link.py:459)


def user_nodes(labels, unseparated=False):
    """src/main_link.py:456-459: all nodes, or those whose id does not start with '9999999'."""
    labels = np.asarray(labels, dtype=np.int64)
    if unseparated:
        return labels
    keep = np.array([not str(int(x)).startswith(ITEM_PREFIX) for x in labels], dtype=bool)
    return labels[keep]


def add_edges(vectors, mode="ratio", ratio=0.1, thre=0.5, block_rows=4096, sim_method="cos"):
    """vectors: float [n, d] device tensor of the user nodes, in user_nodes order.
    Returns (src_idx, dst_idx, weight) tensors over user indices, in the reference's output order
    (user by user; inside a user: ranking order for "ratio", list order otherwise)."""
    n = int(vectors.shape[0])
    X = simsel.prepare(vectors, sim_method)
    dev = X.device
    if mode == "relu-ratio":
        mode, thre = "relu", ratio
    if mode not in ("ratio", "step", "relu", "linear"):
        raise ValueError("user-edges-mode value fault: " + str(mode))
    k = int(n * ratio)
    srcs, dsts, ws = [], [], []
    block_rows = max(1, min(int(block_rows), (1 << 31) // max(n, 1)))   # score block <= 8 GiB
    for b in range(0, n, block_rows):
        e = min(n, b + block_rows)
        sim = simsel.score_block(X, b, e - b, X, sim_method, zero_diag_off=0)   # user_user_sim_list[i] = 0
        rows = torch.arange(b, e, device=dev)
        if mode == "ratio":
            if k == 0:
                continue
            cols, _ = simsel.rows_topk(sim, n, k)
            srcs.append(rows[:, None].expand(-1, k).reshape(-1))
            dsts.append(cols.reshape(-1).long())
            ws.append(torch.ones((e - b) * k, dtype=torch.float32, device=dev))
        elif mode in ("step", "relu"):
            r, c, v = simsel.rows_above(sim, n, thre)                # row-major = user by user, list order
            srcs.append(r + b)
            dsts.append(c.long())
            ws.append(torch.ones(r.numel(), dtype=torch.float32, device=dev) if mode == "step" else v)
        else:
            srcs.append(rows[:, None].expand(-1, n).reshape(-1))
            dsts.append(torch.arange(n, device=dev)[None, :].expand(e - b, -1).reshape(-1))
            ws.append(sim.reshape(-1))
    if not srcs:
        z = torch.zeros(0, dtype=torch.int64, device=dev)
        return z, z.clone(), torch.zeros(0, dtype=torch.float32, device=dev)
    return torch.cat(srcs), torch.cat(dsts), torch.cat(ws)


def add_weighted_edges(graph, src, dst, w):
    """networkx ``G.add_weighted_edges_from`` on the (undirected or directed) CsrGraph: edges are
    applied in order, a repeated pair keeps its LAST weight, existing edges get the new weight,
    unweighted graphs become weighted (old edges keep weight 1).  Labels must exist already."""
    from .csr import CsrGraph
    src = np.asarray(src, dtype=np.int64)
    dst = np.asarray(dst, dtype=np.int64)
    w = np.asarray(w, dtype=np.float64)
    n = graph.n_nodes
    s_old = graph.src_of().astype(np.int64)
    d_old = graph.col.astype(np.int64)
    w_old = np.ones(len(d_old)) if graph.w is None else graph.w
    s_new, d_new = graph.dense_of(src).astype(np.int64), graph.dense_of(dst).astype(np.int64)
    if graph.directed:
        s_all, d_all, w_all = np.concatenate([s_old, s_new]), np.concatenate([d_old, d_new]), np.concatenate([w_old, w])
        key = s_all * n + d_all
    else:
        keep = s_old <= d_old                                    # one entry per undirected pair
        s_all = np.concatenate([s_old[keep], np.minimum(s_new, d_new)])
        d_all = np.concatenate([d_old[keep], np.maximum(s_new, d_new)])
        w_all = np.concatenate([w_old[keep], w])
        key = s_all * n + d_all
    m = len(key)
    _, last_rev = np.unique(key[::-1], return_index=True)        # last occurrence of every pair wins
    win = m - 1 - last_rev
    a, b, ww = s_all[win], d_all[win], w_all[win]
    if not graph.directed:
        loop = a == b
        a, b, ww = np.concatenate([a, b[~loop]]), np.concatenate([b, a[~loop]]), np.concatenate([ww, ww[~loop]])
    perm = np.lexsort((b, a))
    a, b, ww = a[perm], b[perm], ww[perm]
    row_ptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(np.bincount(a, minlength=n), out=row_ptr[1:])
    return CsrGraph(graph.labels, row_ptr, b.astype(np.int32), ww, graph.start_order, graph.directed)


def augment_graph(graph, vectors_by_dense, mode="ratio", ratio=0.1, thre=0.5, unseparated=False, sim_method="cos"):
    """src/main_link.py:568-573: user nodes -> similarity edges -> nx_G.add_weighted_edges_from.
    vectors_by_dense: float [N, d] tensor (row = dense node id).  Returns (new CsrGraph, #edges added)."""
    users = user_nodes(graph.labels[graph.start_order], unseparated)      # g.nodes() order
    u_dense = torch.as_tensor(graph.dense_of(users).astype(np.int64), device=vectors_by_dense.device)
    s, d, w = add_edges(vectors_by_dense[u_dense], mode, ratio, thre, sim_method=sim_method)
    s, d, w = s.cpu().numpy(), d.cpu().numpy(), w.cpu().numpy().astype(np.float64)
    return add_weighted_edges(graph, users[s], users[d], w), len(s)
