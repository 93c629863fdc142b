"""Where does preprocess_transition_probs spend its wall time on C3 (BENCH_r01: 2.76 s, kernels 0.37 s)?
Phase timers with a device sync after each phase.  Usage: python tools/preprocess_probe.py [C3|C2]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch

import node2vec
from n2v_hip import synth

key = sys.argv[1] if len(sys.argv) > 1 else "C3"
gkey, p, q = {"C3": ("C3", 0.25, 4.0), "C2": ("C2", 1.0, 1.0)}[key]
t0 = time.perf_counter()
cg, info = synth.make_config_graph(gkey)
print("graph %.2fs" % (time.perf_counter() - t0), flush=True)
os.environ["N2V_TIMING"] = "1"
for rep in range(2):
    g = node2vec.Graph.from_csr(cg, p, q, device="cuda:0", rng="philox", seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.preprocess_transition_probs()
    torch.cuda.synchronize()
    print("rep %d: preprocess %.3fs  phases %s" % (rep, time.perf_counter() - t0,
          {k: round(v, 3) for k, v in g._engine.timings.items()}), flush=True)
    print("   reserved %.1f GB allocated %.1f GB" % (torch.cuda.memory_reserved() / 1e9, torch.cuda.memory_allocated() / 1e9), flush=True)
    del g
    torch.cuda.empty_cache()
