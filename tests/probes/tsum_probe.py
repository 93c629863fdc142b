"""Product path of merge="tsum" (n2v_hip.sgns.train_simulated_replicas) on the probe graphs: wire dtype and theta."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, p)
import numpy as np
import torch

import node2vec
from n2v_hip import linkpred, sgns
import replica_auc_probe as rap

kind = sys.argv[1] if len(sys.argv) > 1 else "pp"
seq = {"pp": 0.89607, "hub": 0.86678}.get(kind)
g, te, neg = rap.setup(kind)
Gr = node2vec.Graph.from_csr(g, 1.0, 1.0, rng="philox", seed=1)
Gr.preprocess_transition_probs()
rounds = 10
corpus = Gr.simulate_walks(rounds, 80)
te_d = np.stack([g.dense_of(te[:, 0]), g.dense_of(te[:, 1])], 1)
neg_d = np.stack([g.dense_of(neg[:, 0]), g.dense_of(neg[:, 1])], 1)
n = g.n_nodes
counts = torch.bincount(corpus.walks.reshape(-1).long(), minlength=n)
for G in [int(x) for x in os.environ.get("GS", "8").split(",")]:
    for wire in (torch.bfloat16,):
        for theta in [float(x) for x in os.environ.get("THETAS", "500,125").split(",")]:
            for budget in [float(x) for x in os.environ.get("BUDGETS", "24").split(",")]:
                sgns.TSUM_THETA, sgns.TSUM_STALENESS_BUDGET = theta, budget
                models, shards = [], []
                for r in range(G):
                    m = sgns.SgnsModel(n, dim=128, window=10, negative=5, seed=1)
                    m.build_vocab(counts=counts)
                    models.append(m)
                    b, e = sgns.shard_bounds(n, G, r)
                    idx = (torch.arange(rounds, device="cuda")[:, None] * n + torch.arange(b, e, device="cuda")[None, :]).reshape(-1)
                    shards.append((corpus.walks[idx].contiguous(), corpus.lens[idx].contiguous(), b * rounds))
                t = time.time()
                # SumTierPlan reads the module constants at call time through its defaults -> pass explicitly
                orig = sgns.SumTierPlan.__init__.__defaults__
                sgns.SumTierPlan.__init__.__defaults__ = (theta, int(os.environ.get("TIERS", orig[1]))) + orig[2:]
                ns = sgns.train_simulated_replicas(models, shards, n_walks_global=corpus.walks.shape[0], merge="tsum",
                                                   wire_dtype=wire)
                sgns.SumTierPlan.__init__.__defaults__ = orig
                torch.cuda.synchronize()
                auc = linkpred.get_roc_score(models[0].vectors(), te_d, neg_d)[0]
                print("[%s] G=%d tiers=%s theta=%g budget=%g base syncs=%d: AUC %.5f (%+.5f) %.0fs" % (
                    kind, G, os.environ.get("TIERS", "4"), theta, budget, ns, auc, auc - seq, time.time() - t), flush=True)
