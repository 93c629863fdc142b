"""Probe: where the reference-exact walk (rng="numpy") and preprocess_transition_probs spend their time on C3.
Run plain for wall times, or under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import numpy as np
import torch

import node2vec
from n2v_hip import synth


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    cg, info = synth.make_config_graph(name)
    p, q = (1.0, 1.0) if name == "C2" else (0.25, 4.0)
    os.environ["N2V_TIMING"] = "1"
    g = node2vec.Graph.from_csr(cg, p, q, rng="numpy")
    for i in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        g.preprocess_transition_probs()
        torch.cuda.synchronize()
        print("preprocess #%d: %.3f s  phases %s" % (i, time.perf_counter() - t,
              {k: round(v, 4) for k, v in g._engine.timings.items()}), flush=True)
    os.environ.pop("N2V_TIMING")
    np.random.seed(5)
    g.simulate_walks(1, 80)   # warm-up: jump polynomials of the batch strides, allocator
    ref = None
    variants = [("linear, one chunk", dict(linear_uniforms=True)), ("tiled, one chunk", dict()),
                ("tiled, 5 rounds/chunk", dict(uniform_chunk_rounds=5)), ("tiled, 2 rounds/chunk", dict(uniform_chunk_rounds=2)),
                ("tiled, 1 round/chunk", dict(uniform_chunk_rounds=1))]
    only = os.environ.get("EXACT_VARIANT")
    if only is not None:
        variants = [variants[int(only)]]
    for label, kw in variants:
        for k in ("linear_uniforms", "uniform_chunk_rounds"):
            g.__dict__.pop(k, None)
        g.__dict__.update(kw)
        for rep in range(3):
            np.random.seed(123)
            torch.cuda.synchronize()
            t = time.perf_counter()
            c = g.simulate_walks(rounds, 80)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            steps = int((c.lens.long() - 1).sum().item())
            print("%s rng=numpy [%s], %d rounds: %.4f s  %.3e steps/s end to end" % (name, label, rounds, dt, steps / dt), flush=True)
            h = int(c.walks.long().sum().item())
            if ref is None:
                ref = h
            assert h == ref, "walks differ between variants"
            del c
    for k in ("linear_uniforms", "uniform_chunk_rounds"):
        g.__dict__.pop(k, None)
    print("walk checksum", h)
    g.rng = "philox"
    for rep in range(2):
        torch.cuda.synchronize()
        t = time.perf_counter()
        c = g.simulate_walks(rounds, 80)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("%s rng=philox, %d rounds: %.4f s  %.3e steps/s" % (name, rounds, dt, int((c.lens.long() - 1).sum().item()) / dt), flush=True)
        del c


if __name__ == "__main__":
    main()
