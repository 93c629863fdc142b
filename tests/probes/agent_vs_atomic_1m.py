"""The benchmarked row mode at C3's size, without a 10-hour comparator run: on a 10^6-node hub graph (10 x 80 walks,
d = 128) train once with lossless atomic rows — which the 131k / 400k fixtures put within 1e-4 of the sequential
comparator once sentences are handed out in order — and once with the agent rows `auto` picks, and compare the
link-prediction AUC of the two.  python tests/probes/agent_vs_atomic_1m.py [case]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import band_cases
import node2vec
from n2v_hip import linkpred, sgns

name = sys.argv[1] if len(sys.argv) > 1 else "hub1m_10x80"
t0 = time.time()
case = band_cases.build(name)
g = case["graph"]
print("%s: %d rows, %d training edges, max degree %d (host %.0f s)" % (name, g.n_nodes, g.nnz // 2, int(g.degrees.max()), time.time() - t0), flush=True)
G = node2vec.Graph.from_csr(g, 1.0, 1.0, rng="philox", seed=1)
G.preprocess_transition_probs()
corpus = G.simulate_walks(case["rounds"], case["L"])
counts = torch.bincount(corpus.walks.reshape(-1)[corpus.walks.reshape(-1) >= 0].long(), minlength=g.n_nodes)
del G
res = {}
for mode in ("atomic", "auto", "atomic", "auto"):
    m = sgns.SgnsModel(g.n_nodes, dim=128, window=10, negative=5, seed=1, update_mode=mode)
    m.build_vocab(counts=counts)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    sgns.train(m, corpus.walks, corpus.lens, epochs=1)
    b.record()
    torch.cuda.synchronize()
    auc, ap = linkpred.get_roc_score(m.vectors(), case["te_d"], case["neg_d"])
    res.setdefault(m.update_mode_name, []).append(auc)
    print("%-6s (%s): AUC %.5f AP %.5f  %.2f s  %.3e pairs/s" % (mode, m.update_mode_name, auc, ap, a.elapsed_time(b) / 1e3,
                                                                 m.pairs_trained() / (a.elapsed_time(b) / 1e3)), flush=True)
    del m
for k, v in res.items():
    print("%s: %s" % (k, " ".join("%.5f" % x for x in v)))
if "agent" in res and "atomic" in res:
    print("agent - atomic = %+.5f" % (sum(res["agent"]) / len(res["agent"]) - sum(res["atomic"]) / len(res["atomic"])))
