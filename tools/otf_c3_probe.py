"""On-the-fly walk rate on C2 / C3 (no stored edge tables): steps/s of n2v_walk_on_the_fly."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch

import node2vec
from n2v_hip import synth

for key, n_sub in (("C2", 100000), ("C3", 200000)):
    cg, info = synth.make_config_graph(key)
    eng = node2vec.WalkEngine(cg, 0.25, 4.0, device="cuda:0")
    sub = eng.start_order[:n_sub].contiguous()
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        w, l = eng.walk_on_the_fly(sub, 1, 80, rng="philox", seed=1)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
    print("%s on-the-fly: %d walks x 80 in %.3fs = %.3e steps/s" % (key, n_sub, dt, float((l.long() - 1).sum()) / dt), flush=True)
