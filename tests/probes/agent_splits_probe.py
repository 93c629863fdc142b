"""Does agent-scope row sharing survive walk_splits > 1 (several wavefronts on ONE sentence) under the tiered merges?
8 simulated replicas on the 131 072-node hub graph against the committed sequential-comparator fixture; update_mode
"auto" (atomic rows for split launches — the shipped rule) vs explicit "agent" with allow_out_of_band (agent rows
everywhere).  python tests/probes/agent_splits_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_sgns_band as tb

for G in (8, 2):
    for mode, allow in (("auto", False), ("agent", True)):
        t = time.time()
        auc, cpu, n_syncs, name = tb._simulated_replicas("hub131k_10x80", G, mode, allow)
        print("G=%d %s(%s): AUC %.5f vs %.5f (%+.5f)  base syncs %d  %.0f s" % (G, mode, name, auc, cpu, auc - cpu, n_syncs, time.time() - t), flush=True)
