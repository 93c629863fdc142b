"""Scratch probe (not a test): preprocess + walk throughput on the BASELINE configs."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import numpy as np
import torch

import node2vec
from n2v_hip import synth


def timed(fn, reps=3):
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t)
    return min(ts), out


def main():
    names = sys.argv[1:] or ["C2", "C3"]
    for name in names:
        t = time.perf_counter()
        cg, info = synth.make_config_graph(name)
        print(name, "graph built in %.1fs" % (time.perf_counter() - t), json.dumps(info), flush=True)
        p, q = (1.0, 1.0) if name in ("C2",) else (0.25, 4.0)
        g = node2vec.Graph.from_csr(cg, p, q, rng="philox", seed=1)
        t = time.perf_counter()
        g.preprocess_transition_probs()
        torch.cuda.synchronize()
        print(name, "preprocess %.3fs  slots=%d (%.2f GB) first_order=%s" % (
            time.perf_counter() - t, g._engine.total_slots, g._engine.total_slots * 16 / 1e9,
            g._engine.first_order), flush=True)
        eng = g._engine
        L = 80
        for r in (1, 10):
            dt, (w, l) = timed(lambda: eng.walk(eng.start_order, r, L, rng="philox", seed=1))
            steps = int((l.long() - 1).sum().item())
            print(name, "philox r=%d: %.4fs  %.3e steps/s" % (r, dt, steps / dt), flush=True)
            del w, l
        n = cg.n_nodes
        U = torch.rand(2 * (L - 1) * n, dtype=torch.float64, device=eng.device)
        dt, (w, l) = timed(lambda: eng.walk(eng.start_order, 1, L, rng="uniforms", uniforms=U))
        steps = int((l.long() - 1).sum().item())
        print(name, "uniform-buffer r=1: %.4fs  %.3e steps/s" % (dt, steps / dt), flush=True)
        del g, eng, U, w, l
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
