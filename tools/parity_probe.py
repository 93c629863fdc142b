"""Scratch probe: end-to-end speed of the reference-exact mode (rng="numpy": numpy's global
MT19937 stream) with the uniforms generated on the device vs by numpy on the host."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import numpy as np
import torch

import node2vec
from n2v_hip import mt19937, synth


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C3"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cg, info = synth.make_config_graph(name)
    p, q = (1.0, 1.0) if name == "C2" else (0.25, 4.0)
    g = node2vec.Graph.from_csr(cg, p, q, rng="numpy")
    g.preprocess_transition_probs()
    torch.cuda.synchronize()
    # generator alone
    for n in (10**6, 10**8):
        for streams in (1, 128, 1024, 1024, 4096, 4096, 8192, 8192):
            np.random.seed(1)
            torch.cuda.synchronize()
            t = time.perf_counter()
            u = mt19937.global_uniforms_device(n, "cuda:0", n_streams=streams)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            print("mt19937 device: n=%.0e streams=%3d  %.4fs  %.3e doubles/s" % (n, streams, dt, n / dt), flush=True)
            del u
    res = {}
    g.host_rng = False
    np.random.seed(5)
    g.simulate_walks(1, 80)   # warm-up: jump polynomials of the batch strides, allocator
    for r10 in (10,):
        np.random.seed(123)
        torch.cuda.synchronize()
        t = time.perf_counter()
        c = g.simulate_walks(r10, 80)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print("%s rng=numpy (device uniforms, warm), %d rounds: %.3fs  %.3e steps/s end to end" % (
            name, r10, dt, int((c.lens.long() - 1).sum().item()) / dt), flush=True)
        del c
    for host in (False, True):
        g.host_rng = host
        np.random.seed(123)
        torch.cuda.synchronize()
        t = time.perf_counter()
        c = g.simulate_walks(rounds, 80)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        steps = int((c.lens.long() - 1).sum().item())
        res[host] = c.walks
        print("%s rng=numpy (%s uniforms), %d rounds: %.3fs  %.3e steps/s end to end" % (
            name, "host" if host else "device", rounds, dt, steps / dt), flush=True)
    print("identical walks:", bool(torch.equal(res[False], res[True])))


if __name__ == "__main__":
    main()
