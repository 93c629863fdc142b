"""AUC against the committed sequential-comparator fixture as a function of the SGNS grid (wavefronts in flight =
staleness of the racing updates), per row-sharing mode.  python tests/probes/grid_band_probe.py [case]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_gpu_sgns_band as tb
from n2v_hip import linkpred, sgns

name = sys.argv[1] if len(sys.argv) > 1 else "hub131k_10x80"
g, corpus, counts, te_d, neg_d, fx = tb.gpu_case(name)
print("%s: %d rows, comparator %.5f" % (name, g.n_nodes, fx["auc_cpu"]), flush=True)
for mode in ("atomic", "agent"):
    for blocks in (0, 2048, 1536, 1024, 768, 512, 256, 128):
        aucs = []
        for rep in range(2):
            m = sgns.SgnsModel(g.n_nodes, dim=128, window=10, negative=5, seed=1, update_mode=mode, allow_out_of_band=True)
            m.build_vocab(counts=counts)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            m.train_pass(corpus.walks, corpus.lens, sentences_base=0, sentences_total=corpus.walks.shape[0], walk_id_base=0,
                         max_blocks=blocks)
            b.record()
            torch.cuda.synchronize()
            aucs.append(linkpred.get_roc_score(m.vectors(), te_d, neg_d)[0])
        print("%-6s grid %4s: AUC %s  vs comparator %+.5f %+.5f   %.2f s (%.3e pairs/s)" % (
            mode, blocks or "dflt", " ".join("%.5f" % x for x in aucs), aucs[0] - fx["auc_cpu"], aucs[1] - fx["auc_cpu"],
            a.elapsed_time(b) / 1e3, m.pairs_trained() / (a.elapsed_time(b) / 1e3)), flush=True)
