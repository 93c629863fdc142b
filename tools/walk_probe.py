"""Scratch probe (not a test): preprocess + walk throughput on the BASELINE configs, for the
thin (16-B slots + records) and fat (32-B slots) table layouts and the on-the-fly kernel."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
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
        eng = node2vec.WalkEngine(cg, p, q)
        g._engine = eng
        torch.cuda.synchronize()
        t = time.perf_counter()
        eng.preprocess(fat=False)
        torch.cuda.synchronize()
        t_thin = time.perf_counter() - t
        t = time.perf_counter()
        eng.build_fat()
        torch.cuda.synchronize()
        print(name, "preprocess thin %.3fs + fat expansion %.3fs  slots=%d (thin %.1f GB, fat %.1f GB) first_order=%s" % (
            t_thin, time.perf_counter() - t, eng.total_slots, eng.total_slots * 16 / 1e9, eng.total_slots * 32 / 1e9,
            eng.first_order), flush=True)
        L = 80
        for layout in ("thin", "fat"):
            for r in (1, 10):
                dt, (w, l) = timed(lambda: eng.walk(eng.start_order, r, L, rng="philox", seed=1, layout=layout))
                steps = int((l.long() - 1).sum().item())
                print(name, "%-4s philox r=%d: %.4fs  %.3e steps/s" % (layout, r, dt, steps / dt), flush=True)
                del w, l
            n = cg.n_nodes
            U = torch.rand(2 * (L - 1) * n, dtype=torch.float64, device=eng.device)
            dt, (w, l) = timed(lambda: eng.walk(eng.start_order, 1, L, rng="uniforms", uniforms=U, layout=layout))
            steps = int((l.long() - 1).sum().item())
            print(name, "%-4s uniform-buffer r=1: %.4fs  %.3e steps/s" % (layout, dt, steps / dt), flush=True)
            del U, w, l
        ns = min(cg.n_nodes, 200000)
        dt, (w, l) = timed(lambda: eng.walk_on_the_fly(eng.start_order[:ns].contiguous(), 1, L, rng="philox", seed=1), reps=2)
        steps = int((l.long() - 1).sum().item())
        print(name, "on-the-fly kernel, %d starts: %.4fs  %.3e steps/s" % (ns, dt, steps / dt), flush=True)
        del g, eng, w, l
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
