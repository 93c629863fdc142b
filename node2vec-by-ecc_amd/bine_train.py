"""Drop-in for the training surface of the reference's src/bine_train.py on MI355X.

`train(args, gul)` (src/bine_train.py:433-515), `walk_generator` (:210-222), `get_context_and_negative_samples`
(:225-240), `save_to_file` / `ndarray_tostring` (:519-531), `top_N` and its metrics (:311-406) and the argument
defaults of `parser()` (:533-597).  The per-vertex dicts the reference returns
(`node_list_u[u]['embedding_vectors']`, shape (1, d)) are views over the device tables copied back once.
`add_user_edge` + `run` reproduce `main()`'s second stage (:614-622): user-user similarity edges (all-pairs
cosine of the user embeddings, selection by ratio / step / relu / relu-ratio / linear on the device,
n2v_hip/augment.py) are added to the graph and `train` runs again — in the reference that changes the HITS input only
(edge_list and the projections are not rebuilt there), and so it does here.
"""
import argparse
import math
import os

import numpy as np
import torch

from bine_graph_utils import GraphUtils  # noqa: F401  (re-export, as the reference module does)


def default_args(**overrides):
    """The reference's argparse defaults (src/bine_train.py:538-595)."""
    d = dict(train_data=r"BiNE/data/test_rating_train.dat", test_data=r"BiNE/data/test_rating_test.dat",
             model_name="dblp", ws=5, ns=4, d=128, maxT=32, minT=1, p=0.15, alpha=0.01, beta=0.01, gamma=0.1, lam=0.01,
             max_iter=50, top_n=10, rec=1, large=1, add_user_edges=True, user_edges_mode="ratio",
             user_edges_ratio=0.1, user_edges_thre=0.5, sim_method="cos", verbose=False)
    d.update(overrides)
    return argparse.Namespace(**d)


def walk_generator(gul, args):
    gul.calculate_centrality()
    gul.homogeneous_graph_random_walks_for_large_bipartite_graph(datafile=args.train_data, percentage=args.p,
                                                                 maxT=args.maxT, minT=args.minT)
    return gul


def get_context_and_negative_samples(gul, args):
    gul.get_negs()
    gul.get_context_and_negatives()
    return gul


class _NodeList(dict):
    """node_list_u / node_list_v: label -> {'embedding_vectors': (1, d), 'context_vectors': (1, d)}."""

    def __init__(self, labels, emb, ctx):
        super().__init__()
        for i, lab in enumerate(labels):
            self[lab] = {"embedding_vectors": emb[i:i + 1], "context_vectors": ctx[i:i + 1]}


def train(args, gul, mode="parallel"):
    """src/bine_train.py:433-515.  Returns (node_list_u, roc, ap) like the reference (roc = ap = 0 there too);
    `train.last` keeps node_list_v, the losses and the final learning rate."""
    walk_generator(gul, args)
    get_context_and_negative_samples(gul, args)
    eng = gul.engine
    eng.init_embeddings(args.d)
    losses = eng.train(max_iter=args.max_iter, alpha=args.alpha, beta=args.beta, gamma=args.gamma, lam=args.lam,
                       ws=args.ws, ns=args.ns, mode=mode)
    g = gul.graph
    node_list_u = _NodeList(g.user_labels.tolist(), eng.vectors("u"), eng.vectors("u", "context"))
    node_list_v = _NodeList(g.item_labels.tolist(), eng.vectors("v"), eng.vectors("v", "context"))
    train.last = {"node_list_v": node_list_v, "losses": losses, "lam": eng.lam}
    model_path = os.path.join("../", args.model_name) if getattr(args, "model_path", None) is None else args.model_path
    if getattr(args, "save", False):
        os.makedirs(model_path, exist_ok=True)
        save_to_file(node_list_u, node_list_v, model_path, args)
    if args.rec and getattr(args, "test_rates", None) is not None:
        test_user, test_item, test_rate = args.test_rates
        train.last["metrics"] = top_N(test_user, test_item, test_rate, node_list_u, node_list_v, args.top_n)
    return node_list_u, 0, 0


train.last = {}


def add_user_edge(args, gul, sim_method="cos", by_matrix=True):
    """src/bine_train.py:160-181 + :620: similarity edges between users, added to the graph hits() sees.
    Returns the number of distinct user pairs added.  sim_method 'cos' / 'pearson' / 'jsd' as :55-71 (all three
    on the device, csrc/n2v_sim.hip; the scores are formed in fp32)."""
    if sim_method not in ("cos", "pearson", "jsd"):
        raise ValueError("sim_method %r" % (sim_method,))
    from n2v_hip import augment
    eng = gul.engine
    vec = eng.emb[: gul.graph.n_u, : eng.dim]
    s, d, w = augment.add_edges(vec, args.user_edges_mode, args.user_edges_ratio, args.user_edges_thre,
                                sim_method=sim_method)
    return eng.add_user_edges(s.cpu().numpy(), d.cpu().numpy(), w.cpu().numpy().astype(np.float64))


def run(args, gul, mode="parallel"):
    """main() after construct_training_graph (src/bine_train.py:612-622): train, then — with args.add_user_edges —
    add the similarity edges and train again."""
    node_list_u, roc, ap = train(args, gul, mode)
    first = dict(train.last)
    if args.add_user_edges:
        n_added = add_user_edge(args, gul, sim_method=args.sim_method, by_matrix=False)
        node_list_u, roc, ap = train(args, gul, mode)
        train.last["user_edges_added"] = n_added
        train.last["first_stage"] = first
    return node_list_u, roc, ap


def ndarray_tostring(array):
    string = ""
    for item in array[0]:
        string += str(item).strip() + " "
    return string + "\n"


def save_to_file(node_list_u, node_list_v, model_path, args):
    with open(os.path.join(model_path, "vectors_u.dat"), "w") as fw_u:
        for u in node_list_u.keys():
            fw_u.write(str(u) + " " + ndarray_tostring(node_list_u[u]["embedding_vectors"]))
    with open(os.path.join(model_path, "vectors_v.dat"), "w") as fw_v:
        for v in node_list_v.keys():
            fw_v.write(str(v) + " " + ndarray_tostring(node_list_v[v]["embedding_vectors"]))


def read_data(filename):
    """DataUtils.read_data (src/bine_data_utils.py:56-70): users, items, rates[user][item]."""
    users, items, rates = set(), set(), {}
    with open(filename, "r", encoding="UTF-8") as fin:
        for line in fin:
            if not line.strip():
                continue
            user, item, rate = line.strip().split()
            rates.setdefault(user, {})[item] = float(rate)
            users.add(user)
            items.add(item)
    return users, items, rates


def top_N(test_u, test_v, test_rate, node_list_u, node_list_v, top_n):
    """src/bine_train.py:311-359: score every (test user, test item) by U.V (0 for unknown vertices), recommend the
    top_n items per user, average F1 / MAP / MRR / NDCG against the user's test items.  The score matrix is one
    library GEMM on the device; ranking ties follow torch.topk instead of Python's sort."""
    test_u, test_v = list(test_u), list(test_v)
    d = next(iter(node_list_u.values()))["embedding_vectors"].shape[1] if len(node_list_u) else 1
    dev = "cuda" if torch.cuda.is_available() else "cpu"

    def stack(labels, table):
        m = np.zeros((len(labels), d))
        for i, x in enumerate(labels):
            if x in table:
                m[i] = table[x]["embedding_vectors"][0]
        return torch.from_numpy(m).to(dev)

    scores = stack(test_u, node_list_u) @ stack(test_v, node_list_v).T
    k = min(len(test_v), top_n)
    top = torch.topk(scores, k, dim=1).indices.cpu().numpy()
    precision_list, recall_list, ap_list, ndcg_list, rr_list = [], [], [], [], []
    for i, u in enumerate(test_u):
        ranked = [test_v[j] for j in top[i]]
        truth = [it for it, _ in sorted(test_rate[u].items(), key=lambda kv: -kv[1])]
        pre, rec = precision_and_racall(ranked, truth)
        precision_list.append(pre)
        recall_list.append(rec)
        ap_list.append(AP(ranked, truth))
        rr_list.append(RR(ranked, truth))
        ndcg_list.append(nDCG(ranked, truth))
    precison = sum(precision_list) / len(precision_list)
    recall = sum(recall_list) / len(recall_list)
    f1 = 2 * precison * recall / (precison + recall) if precison + recall > 0 else 0.0
    return f1, sum(ap_list) / len(ap_list), sum(rr_list) / len(rr_list), sum(ndcg_list) / len(ndcg_list)


def nDCG(ranked_list, ground_truth):
    dcg = 0
    idcg = IDCG(len(ground_truth))
    for i in range(len(ranked_list)):
        if ranked_list[i] not in ground_truth:
            continue
        dcg += 1 / math.log(i + 2, 2)
    return dcg / idcg


def IDCG(n):
    return sum(1 / math.log(i + 2, 2) for i in range(n))


def AP(ranked_list, ground_truth):
    hits, sum_precs = 0, 0.0
    for i in range(len(ranked_list)):
        if ranked_list[i] in ground_truth:
            hits += 1
            sum_precs += hits / (i + 1.0)
    return sum_precs / len(ground_truth) if hits > 0 else 0.0


def RR(ranked_list, ground_list):
    for i in range(len(ranked_list)):
        if ranked_list[i] in ground_list:
            return 1 / (i + 1.0)
    return 0


def precision_and_racall(ranked_list, ground_list):
    hits = sum(1 for x in ranked_list if x in ground_list)
    return hits / (1.0 * len(ranked_list)), hits / (1.0 * len(ground_list))
