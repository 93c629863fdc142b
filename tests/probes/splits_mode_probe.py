"""Tiered merges, 8 simulated replicas on the 131k hub graph: AUC against the committed comparator and wall time of the
simulation for (row-sharing mode) x (wavefronts per sentence in the short launches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_gpu_sgns_band as tb
from n2v_hip import linkpred, sgns

name = sys.argv[1] if len(sys.argv) > 1 else "hub131k_10x80"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g, corpus, counts, te_d, neg_d, fx = tb.gpu_case(name)
n, rounds = g.n_nodes, fx["rounds"]
for mode in ("atomic", "agent"):
    for S in (1, 2, 4, 8, 16, 80):
        models, shards = [], []
        for r in range(G):
            m = sgns.SgnsModel(n, dim=128, window=10, negative=5, seed=1, update_mode=mode, allow_out_of_band=True)
            m.build_vocab(counts=counts)
            models.append(m)
            b, e = sgns.shard_bounds(n, G, r)
            idx = (torch.arange(rounds, device="cuda")[:, None] * n + torch.arange(b, e, device="cuda")[None, :]).reshape(-1)
            shards.append((corpus.walks[idx].contiguous(), corpus.lens[idx].contiguous(), b * rounds))
        torch.cuda.synchronize(); t = time.time()
        sgns.train_simulated_replicas(models, shards, n_walks_global=corpus.walks.shape[0], merge="tsum", splits=S)
        torch.cuda.synchronize(); dt = time.time() - t
        auc = linkpred.get_roc_score(models[0].vectors(), te_d, neg_d)[0]
        print("%s G=%d %-6s splits %2d: AUC %+.5f vs comparator   %.1f s for all replicas" % (name, G, mode, S, auc - fx["auc_cpu"], dt), flush=True)
        del models, shards
