"""Scratch probe: cost of one on-the-fly step as a function of the table size K (star graphs:
every second step of a walk stands on the hub and rebuilds a K-slot table)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import numpy as np
import torch

import node2vec
from n2v_hip import csr

for D in (64, 256, 500, 600, 2000, 16000):
    # 64 hubs so that concurrent waves do not all touch the same rows; each hub has D leaves
    hubs = 64
    src = np.repeat(np.arange(hubs), D)
    dst = hubs + np.arange(hubs * D)
    cg = csr.from_edges(src, dst, None, False)
    eng = node2vec.WalkEngine(cg, 0.25, 4.0)
    starts = torch.from_numpy(cg.dense_of(dst[:: max(1, (hubs * D) // 5120)][:5120]).astype(np.int32)).cuda()
    L = 41
    eng.walk_on_the_fly(starts, 1, L, rng="philox", seed=1)
    torch.cuda.synchronize()
    t = time.perf_counter()
    w, l = eng.walk_on_the_fly(starts, 1, L, rng="philox", seed=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    hub_steps = starts.numel() * (L - 1) // 2
    waves = min(5120, starts.numel())
    per_step = dt / ((L - 1) // 2) * 1e6 * (waves / starts.numel())
    print("K=%5d: %.4fs for %d walks  -> %.1f us per hub step per wave, %.1f ns per slot" % (
        D, dt, starts.numel(), per_step, per_step * 1e3 / D), flush=True)
