"""A/B of the fat-table walk kernel variants (N2V_WALK_VARIANT, csrc/n2v_walk_fat.hip) on C3 / C2: identical walks,
HIP-event time per launch.  Also times preprocess phases.  Usage (with a lab library, tools/lab/README.md): N2V_HIP_LIB=tools/lab/libn2v_hip_lab.so python tools/lab/walk_variant_probe.py [C3] [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch

import node2vec
from n2v_hip import synth

key = sys.argv[1] if len(sys.argv) > 1 else "C3"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p, q = {"C3": (0.25, 4.0), "C2": (0.25, 4.0)}[key]
cg, info = synth.make_config_graph(key)
os.environ["N2V_TIMING"] = "1"
g = node2vec.Graph.from_csr(cg, p, q, device="cuda:0", rng="philox", seed=1)
t0 = time.perf_counter()
g.preprocess_transition_probs()
torch.cuda.synchronize()
eng = g._engine
print("%s preprocess %.3fs phases %s" % (key, time.perf_counter() - t0, {k: round(v, 3) for k, v in eng.timings.items()}), flush=True)
L = 80
n = cg.n_nodes
walks = torch.empty((n * rounds, L), dtype=torch.int32, device="cuda:0")
lens = torch.empty(n * rounds, dtype=torch.int32, device="cuda:0")
ref = None
for variant in (0, 2, 4, 2, 4):
    os.environ["N2V_WALK_VARIANT"] = str(variant)
    best = 1e9
    for rep in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        eng.walk(eng.start_order, rounds, L, rng="philox", seed=7, out=(walks, lens))
        b.record()
        torch.cuda.synchronize()
        if rep:
            best = min(best, a.elapsed_time(b))
    steps = int((lens.long() - 1).sum().item())
    if ref is None:
        ref = (walks.clone(), lens.clone())
    same = torch.equal(walks, ref[0]) and torch.equal(lens, ref[1])
    print("%s variant %d: %.3f ms  %.3e steps/s  frac(36B/8TB/s) %.3f  identical=%s" % (
        key, variant, best, steps / best * 1e3, steps / best * 1e3 * 36 / 8e12, same), flush=True)
    assert same
del os.environ["N2V_WALK_VARIANT"]
