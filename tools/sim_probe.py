"""Timing of the all-pairs similarity + selection kernels (csrc/n2v_sim.hip): the global top-k scan of
link_prediction (src/main_link.py:123-170) over users x items, and the user x user edge selection
(src/main_link.py:379-453), on the MFMA tile kernel and (N2V_SIM_VECTOR=1) the vector-FMA one."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch

from n2v_hip import augment, simsel


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best, out


g = torch.Generator(device="cuda").manual_seed(0)
for n in (20000, 100000):
    centres = torch.randn(64, 128, device="cuda", generator=g)
    x = centres[torch.randint(0, 64, (n,), device="cuda", generator=g)] + 0.8 * torch.randn(n, 128, device="cuda", generator=g)
    y = centres[torch.randint(0, 64, (n,), device="cuda", generator=g)] + 0.8 * torch.randn(n, 128, device="cuda", generator=g)
    ref = None
    for vec in ("0", "1"):
        os.environ["N2V_SIM_VECTOR"] = vec
        A, B = simsel.prepare(x, "cos"), simsel.prepare(y, "cos")
        dt, (s, r, c) = timed(lambda: simsel.global_topk(A, B, 1000))
        flops = 2.0 * n * n * 128
        print("n=%d %s: global top-1000 of %.1e pairs in %.3fs = %.3e pairs/s (%.1f TFLOP/s incl. selection)" % (
            n, "vector" if vec == "1" else "mfma  ", float(n) * n, dt, n * n / dt, flops / dt / 1e12), flush=True)
        if ref is None:
            ref = (s.clone(), r.clone(), c.clone())
        else:
            same = len(set(zip(r.tolist(), c.tolist())) ^ set(zip(ref[1].tolist(), ref[2].tolist())))
            print("   top-1000 sets differ in %d pairs between the two kernels; max |score diff| %.2e" % (
                same, float((s - ref[0]).abs().max())), flush=True)
        nb = min(n, 8192)
        dt, _ = timed(lambda: simsel.score_block(A, 0, nb, B, "cos"))
        print("   score block %d x %d: %.4fs = %.1f TFLOP/s" % (nb, n, dt, 2.0 * nb * n * 128 / dt / 1e12), flush=True)
    os.environ["N2V_SIM_VECTOR"] = "0"
    for mode, kw in (("ratio", dict(ratio=0.001)), ("step", dict(thre=0.3)), ("relu", dict(thre=0.3))):
        for sim in ("cos", "pearson"):
            dt, (s, d, w) = timed(lambda: augment.add_edges(x, mode, sim_method=sim, **kw), reps=2)
            print("n=%d add_edges %s/%s: %.3fs  %.3e user pairs/s  (%d edges)" % (n, mode, sim, dt, n * n / dt, s.numel()), flush=True)
