"""Pure on-the-fly and 1/3-budget hybrid walk rate on C3 for the library named by N2V_HIP_LIB (A/B builds of tools/lab/otf_variants.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch
import node2vec
from n2v_hip import synth

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
p, q = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.25, 4.0)
cg, info = synth.make_config_graph(name)
g = node2vec.Graph.from_csr(cg, p, q, device="cuda:0", rng="philox", seed=1)
eng = g._graph_engine()
n_walks = int(os.environ.get("N_WALKS", 400000))
starts = eng.start_order[:n_walks].contiguous()


def rate(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        w, l = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return float((l.long() - 1).sum().item()) / best, int(w.long().sum().item())

r, chk = rate(lambda: eng.walk_on_the_fly(starts, 1, 80, rng="philox", seed=7))
print("%s p=%g q=%g lib=%s on-the-fly: %.3e steps/s (checksum %d)" % (name, p, q, os.path.basename(os.environ.get("N2V_HIP_LIB", "product")), r, chk), flush=True)
if os.environ.get("HYBRID", "1") == "1":
    full = eng.total_edge_slots * 32
    eng.preprocess(budget_bytes=int(full / 3))
    r, chk2 = rate(lambda: eng.walk(starts, 1, 80, rng="philox", seed=7))
    print("   1/3 budget hybrid: %.3e steps/s identical %s" % (r, chk2 == chk), flush=True)
