import os, sys
"""AUC against the 400k-node comparator fixture for SGNS grids around the default: even and uneven numbers of workgroups
per CU (DESIGN.md 3).  GRIDS="0 1561 1536 3072" python tests/probes/grid_resonance_probe.py; N2V_HIP_LIB selects a lab build."""
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import test_gpu_sgns_band as tb
from n2v_hip import linkpred, sgns
name = "hub400k_10x80"
g, corpus, counts, te_d, neg_d, fx = tb.gpu_case(name)
print("lib %s" % os.path.basename(os.environ.get("N2V_HIP_LIB", "product")), flush=True)
print("%s: %d rows, comparator %.5f, walks %s" % (name, g.n_nodes, fx["auc_cpu"], tuple(corpus.walks.shape)), flush=True)
for mode in ("agent", "atomic"):
    for blocks in [int(x) for x in os.environ.get("GRIDS", "0 1561 1562 1560 1536 1600 3072").split()]:
        m = sgns.SgnsModel(g.n_nodes, dim=128, window=10, negative=5, seed=1, update_mode=mode, allow_out_of_band=True)
        m.build_vocab(counts=counts)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        m.train_pass(corpus.walks, corpus.lens, sentences_base=0, sentences_total=corpus.walks.shape[0], walk_id_base=0, max_blocks=blocks)
        b.record()
        torch.cuda.synchronize()
        auc = linkpred.get_roc_score(m.vectors(), te_d, neg_d)[0]
        print("%-6s grid %4s: AUC %.5f (%+.5f)  %.2f s" % (mode, blocks or "dflt", auc, auc - fx["auc_cpu"], a.elapsed_time(b) / 1e3), flush=True)
