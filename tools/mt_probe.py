"""Scratch probe: cost of the pieces of mt19937.global_uniforms_device (device jump, fill, host state advance)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import numpy as np, torch
from n2v_hip import mt19937, _lib

def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

key = np.random.RandomState(3).get_state()[1]
n = 10**8
for ns in (64, 256, 1024, 4096):
    wps = -(-2 * n // (ns * 624)) * 624
    ms_jump = t(lambda: mt19937.jump_states_device(key, wps, ns, "cuda:0"))
    st = mt19937.jump_states_device(key, wps, ns, "cuda:0")
    out = torch.empty(n, dtype=torch.float64, device="cuda:0")
    lib = _lib.load()
    ms_fill = t(lambda: _lib.check(lib.n2v_mt19937_fill(_lib.ptr(st), ns, 0, wps, n, _lib.ptr(out), None, _lib.stream_ptr(torch.device("cuda:0")))))
    np.random.seed(1)
    ms_adv = t(lambda: mt19937.advance_global_state(n))
    print("streams %5d: jump %.2f ms  fill %.2f ms (%.2e doubles/s)  host advance %.2f ms" % (ns, ms_jump, ms_fill, n / ms_fill * 1e3, ms_adv), flush=True)
