"""How many tiers do the tiered pure-sum merges need to stay inside the +-0.002 band, scored against the committed
sequential-comparator fixtures (simulated replicas, product kernels)?  Fewer tiers = fewer, larger training launches
between merges (4 tiers: 64 sub-intervals per base interval, 3: 16, 2: 4).
python tests/probes/tsum_tiers_probe.py [case ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_sgns_band as tb
from n2v_hip import merge, sgns

cases = sys.argv[1:] or ["hub131k_10x80", "hub20k_10x80", "uniform3k_10x80"]
plan_cls = merge.SumTierPlan
for name in cases:
    for tiers, theta in [tuple(float(x) for x in t.split(":")) for t in os.environ.get("TIERS", "4:125,3:125,3:60,2:125").split(",")]:
        tiers = int(tiers)
        sgns.SumTierPlan = lambda *a, _t=tiers, _th=theta, **k: plan_cls(*a, theta=_th, n_tiers=_t, **k)
        for G in [int(x) for x in os.environ.get("GS", "8,4,2").split(",")]:
            t = time.time()
            auc, cpu, n_syncs, mode = tb._simulated_replicas(name, G, "auto")
            print("%s tiers %d theta %3.0f G=%d: AUC %.5f vs %.5f (%+.5f)  base syncs %d  %.0f s" % (
                name, tiers, theta, G, auc, cpu, auc - cpu, n_syncs, time.time() - t), flush=True)
sgns.SumTierPlan = plan_cls
