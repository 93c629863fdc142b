"""Where does the positive AUC offset of the tiered merges on the 20k hub graph come from — the merges, or the short
launches that deal a sentence to many wavefronts?  One replica trained through the same launch structure without any
merge, and two replicas with unsplit launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_gpu_sgns_band as tb
from n2v_hip import linkpred, sgns

name = sys.argv[1] if len(sys.argv) > 1 else "hub20k_10x80"
g, corpus, counts, te_d, neg_d, fx = tb.gpu_case(name)
W = corpus.walks.shape[0]


def score(m):
    return linkpred.get_roc_score(m.vectors(), te_d, neg_d)[0] - fx["auc_cpu"]


def fresh():
    m = sgns.SgnsModel(g.n_nodes, dim=128, window=10, negative=5, seed=1)
    m.build_vocab(counts=counts)
    return m

for S in (1, 8, 80):
    m = fresh()
    m.train_pass(corpus.walks, corpus.lens, sentences_base=0, sentences_total=W, walk_id_base=0, splits=S)
    print("%s one launch, splits %2d: %+.5f" % (name, S, score(m)), flush=True)
for n_launch in (34, 34 * 64, 234 * 64):
    for S in (1, "auto"):
        m = fresh()
        for c in range(n_launch):
            b, e = c * W // n_launch, (c + 1) * W // n_launch
            if e > b:
                m.train_pass(corpus.walks[b:e], corpus.lens[b:e], sentences_base=b, sentences_total=W, walk_id_base=b, splits=S)
        print("%s %5d launches in corpus order, splits %s: %+.5f" % (name, n_launch, S, score(m)), flush=True)
for G in (2, 8):
    for S in (1, "auto"):
        models, shards = [], []
        n, rounds = g.n_nodes, fx["rounds"]
        for r in range(G):
            m = fresh()
            models.append(m)
            b, e = sgns.shard_bounds(n, G, r)
            idx = (torch.arange(rounds, device="cuda")[:, None] * n + torch.arange(b, e, device="cuda")[None, :]).reshape(-1)
            shards.append((corpus.walks[idx].contiguous(), corpus.lens[idx].contiguous(), b * rounds))
        sgns.train_simulated_replicas(models, shards, n_walks_global=W, merge="tsum", splits=S)
        print("%s tsum G=%d, splits %s: %+.5f" % (name, G, S, score(models[0])), flush=True)
