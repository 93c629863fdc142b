"""Cost of the pieces of the reference-exact walk on C3-sized inputs: jump (device doubling), fill (linear / tiled) by
number of streams and chunk size, HIP-event times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import numpy as np, torch
from n2v_hip import mt19937, _lib

lib = _lib.load()
dev = torch.device("cuda:0")
key = np.random.RandomState(3).get_state()[1]


def ev_ms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

per = 158 * 10**6
for rounds in (1, 3, 10):
    n = per * rounds
    for ns_req in (1024, 2048, 4096, 8192):
        wps = -(-2 * n // (ns_req * 624)) * 624
        blocks = 1 << int(np.ceil(np.log2(wps // 624)))
        wps = 624 * blocks
        ns = -(-2 * n // wps)
        ms_jump = ev_ms(lambda: mt19937.jump_states_device(key, wps, ns, dev))
        st = mt19937.jump_states_device(key, wps, ns, dev)
        out = torch.empty(mt19937.tiled_size(n, 79), dtype=torch.float64, device=dev)
        ms_lin = ev_ms(lambda: _lib.check(lib.n2v_mt19937_fill(_lib.ptr(st), ns, 0, wps, n, _lib.ptr(out), None, _lib.stream_ptr(dev))))
        ms_til = ev_ms(lambda: _lib.check(lib.n2v_mt19937_fill_tiled(_lib.ptr(st), ns, 0, wps, n, 79, _lib.ptr(out), None, _lib.stream_ptr(dev))))
        t0 = time.perf_counter()
        np.random.seed(1); mt19937.advance_global_state(n)
        host = (time.perf_counter() - t0) * 1e3
        print("rounds %2d  streams %5d (blocks/stream %5d): jump %.2f ms  fill linear %.2f ms  tiled %.2f ms (%.2e doubles/s)  host advance %.1f ms" % (
            rounds, ns, blocks, ms_jump, ms_lin, ms_til, n / ms_til * 1e3, host), flush=True)
        del out
