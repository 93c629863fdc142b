"""Scratch probe: edge-augmentation selection rate (pairs of users scored per second)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch

from n2v_hip import augment

for n in (20000, 100000):
    x = torch.randn(n, 128, device="cuda")
    for mode, kw in (("ratio", dict(ratio=0.001)), ("step", dict(thre=0.3)), ("relu", dict(thre=0.3))):
        torch.cuda.synchronize()
        t = time.perf_counter()
        s, d, w = augment.add_edges(x, mode, **kw)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("n=%d mode=%s: %.3fs  %.3e user pairs/s  (%d edges)" % (n, mode, dt, n * n / dt, s.numel()), flush=True)
