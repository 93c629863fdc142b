"""Link-prediction scoring on the device (src/main_link.py:43-61,173-204 of the reference,
``link_method == "cos"``): batched cosine over edge lists and ROC-AUC / average precision by
rank statistics, so a 10^7-edge evaluation does not go through a Python loop."""
import numpy as np
import torch


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
