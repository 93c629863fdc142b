"""Why does the agent-scope row-sharing mode trail the sequential comparator by 0.24 AUC on a 131 072-node hub graph
after 2 rounds of 40 (tests/test_gpu_fullsize.py, first version), when it equals the lossless atomic mode after
10 rounds of 80 (round 1: 0.88018 vs 0.88011 at 200k nodes)?  Same graph, both GPU modes, several grid sizes,
against the CPU comparator (12 threads; 1 thread for the short corpus)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, p)
import numpy as np
import torch

import node2vec
from n2v_hip import csr, linkpred, sgns
from oracle import c_oracle
from replica_auc_probe import _hub_partition

n = int(sys.argv[1]) if len(sys.argv) > 1 else sgns.AUTO_AGENT_MIN_WORDS
edges = _hub_partition(n=n, k=n // 200, m_in=10 * n, m_out=2 * n, seed=2)
tr, te = linkpred.split_edges(edges)
full = csr.from_edges(edges[:, 0], edges[:, 1], None, False)
g = csr.from_edges(tr[:, 0], tr[:, 1], None, False)
if g.n_nodes != full.n_nodes:
    g = linkpred._with_isolated_nodes(g, full)
neg = linkpred.build_neg_samples(full.labels, edges, 0)
te_d = np.stack([g.dense_of(te[:, 0]), g.dense_of(te[:, 1])], 1)
neg_d = np.stack([g.dense_of(neg[:, 0]), g.dense_of(neg[:, 1])], 1)
G = node2vec.Graph.from_csr(g, 1.0, 1.0, rng="philox", seed=1)
G.preprocess_transition_probs()
for rounds, L in ((2, 40), (5, 40), (10, 80)):
    corpus = G.simulate_walks(rounds, L)
    flat = corpus.walks.cpu().numpy().reshape(-1)
    counts = np.bincount(flat[flat >= 0], minlength=g.n_nodes)
    si, cum = sgns.vocab_tables(counts, 1e-3)
    wk, ln = corpus.walks.cpu().numpy(), corpus.lens.cpu().numpy()
    for thr in ((12, 1) if rounds == 2 else (12,)):
        syn0, syn1 = c_oracle.sgns_init(g.n_nodes, 128, 128, 1)
        t = time.time()
        c_oracle.sgns_train(wk, ln, syn0, syn1, 128, 10, 5, si, cum, n_threads=thr)
        auc = linkpred.get_roc_score(torch.from_numpy(syn0).cuda(), te_d, neg_d)[0]
        print("%dx%d CPU %2d threads: AUC %.5f (%.0fs)" % (rounds, L, thr, auc, time.time() - t), flush=True)
    for mode in ("atomic", "agent"):
        for blocks in (0, 768):
            m = sgns.SgnsModel(g.n_nodes, dim=128, window=10, negative=5, seed=1, update_mode=mode)
            m.build_vocab(corpus.walks)
            m.train_pass(corpus.walks, corpus.lens, sentences_base=0, sentences_total=corpus.walks.shape[0],
                         walk_id_base=0, max_blocks=blocks)
            auc = linkpred.get_roc_score(m.vectors(), te_d, neg_d)[0]
            print("%dx%d GPU %-6s grid %4d: AUC %.5f  pairs %d" % (rounds, L, mode, blocks or 3072, auc, m.pairs_trained()), flush=True)
