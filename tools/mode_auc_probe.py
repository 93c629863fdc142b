"""Scratch probe: do the lossy SGNS sharing modes (agent, plain) agree with the lossless atomic
mode on LARGE vocabularies?  (The sequential CPU comparator is out of reach there, but the atomic
mode was validated against it at small sizes, so GPU-vs-GPU AUC on a hub-heavy community graph
tells whether lost updates still matter at 2e5 / 1e6 nodes.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import node2vec
from n2v_hip import csr, linkpred, sgns
from replica_auc_probe import _hub_partition


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    t = time.time()
    edges = _hub_partition(n=n, k=n // 200, m_in=10 * n, m_out=2 * n, seed=1)
    tr, te = linkpred.split_edges(edges)
    full = csr.from_edges(edges[:, 0], edges[:, 1], None, False)
    g = csr.from_edges(tr[:, 0], tr[:, 1], None, False)
    if g.n_nodes != full.n_nodes:
        g = linkpred._with_isolated_nodes(g, full)
    print("graph: %d nodes, %d train edges, max degree %d (%.0fs)" % (g.n_nodes, len(tr), g.degrees.max(), time.time() - t),
          flush=True)
    neg = linkpred.build_neg_samples(full.labels, edges, 0)
    te_d = np.stack([g.dense_of(te[:, 0]), g.dense_of(te[:, 1])], 1)
    neg_d = np.stack([g.dense_of(neg[:, 0]), g.dense_of(neg[:, 1])], 1)
    G = node2vec.Graph.from_csr(g, 1.0, 1.0, rng="philox", seed=1)
    G.preprocess_transition_probs()
    corpus = G.simulate_walks(10, 80)
    for mode in os.environ.get("MODES", "atomic,agent,plain,atomic+share").split(","):
        for seed in (1, 2):
            m = sgns.SgnsModel(g.n_nodes, dim=128, window=10, negative=5, seed=seed, update_mode=mode.split("+")[0],
                               share_negatives=mode.endswith("+share"))
            m.build_vocab(corpus.walks)
            torch.cuda.synchronize()
            t = time.perf_counter()
            sgns.train(m, corpus.walks, corpus.lens, epochs=1)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            auc, ap = linkpred.get_roc_score(m.vectors(), te_d, neg_d)
            print("%-13s seed %d: AUC %.5f AP %.5f  %.2fs %.2e pairs/s" % (mode, seed, auc, ap, dt, m.pairs_trained() / dt),
                  flush=True)
            del m


if __name__ == "__main__":
    main()
