"""Scratch probe: link-prediction AUC of the GPU SGNS vs the CPU comparator as a function of
GPU concurrency (max_blocks), on the planted-partition graph of tests/test_gpu_sgns.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import node2vec
from n2v_hip import linkpred, sgns
from oracle import c_oracle, sgns_oracle
from test_gpu_sgns import _auc_setup


def main():
    g, te, neg = _auc_setup()
    G = node2vec.Graph.from_csr(g, 1.0, 1.0, rng="philox", seed=1)
    G.preprocess_transition_probs()
    corpus = G.simulate_walks(10, 80)
    te_d = np.stack([g.dense_of(te[:, 0]), g.dense_of(te[:, 1])], 1)
    neg_d = np.stack([g.dense_of(neg[:, 0]), g.dense_of(neg[:, 1])], 1)
    n = corpus.walks.shape[0]
    modes = os.environ.get("MODES", "plain,agent,atomic").split(",")
    for mode, blocks in [(mo, int(x)) for mo in modes for x in (sys.argv[1:] or [1, 16, 256, 2048])]:
        for seed in (1, 2):
            m = sgns.SgnsModel(g.n_nodes, dim=128, window=10, negative=5, seed=seed, update_mode=mode.split("+")[0], share_negatives=mode.endswith("+share"))
            m.build_vocab(corpus.walks)
            t = time.perf_counter()
            m.train_pass(corpus.walks, corpus.lens, 0, n, 0, max_blocks=blocks)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            auc, ap = linkpred.get_roc_score(m.vectors(), te_d, neg_d)
            print(mode, "blocks %5d seed %d: AUC %.5f AP %.5f  %.3fs %.2e pairs/s" % (
                blocks, seed, auc, ap, dt, m.pairs_trained() / dt), flush=True)
    m = sgns.SgnsModel(g.n_nodes, dim=128, window=10, negative=5, seed=1)
    m.build_vocab(corpus.walks)
    si, cum = sgns_oracle.vocab_tables(m.counts, 1e-3)
    wh, lh = corpus.walks.cpu().numpy(), corpus.lens.cpu().numpy()
    for thr, seed in ((1, 1),):
        syn0, syn1 = c_oracle.sgns_init(g.n_nodes, 128, 128, seed)
        t = time.perf_counter()
        c_oracle.sgns_train(wh, lh, syn0, syn1, 128, 10, 5, si, cum, seed=seed, n_threads=thr)
        dt = time.perf_counter() - t
        auc, ap = linkpred.get_roc_score(torch.from_numpy(syn0).cuda(), te_d, neg_d)
        print("cpu threads %d seed %d: AUC %.5f AP %.5f  %.1fs" % (thr, seed, auc, ap, dt), flush=True)


if __name__ == "__main__":
    main()
