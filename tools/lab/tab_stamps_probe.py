"""Per-phase shader cycles of edge_tables_wave_kernel on C3 (diagnostic build: tools/lab/tab_stamps.sh).
N2V_HIP_LIB=tools/lab/libn2v_hip_stamps.so python tools/lab/tab_stamps_probe.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch

import node2vec
from n2v_hip import _lib, synth

cg, info = synth.make_config_graph(sys.argv[1] if len(sys.argv) > 1 else "C3")
g = node2vec.Graph.from_csr(cg, 0.25, 4.0, device="cuda:0", rng="philox", seed=1)
lib = C.CDLL(_lib.SO_PATH)
buf = (C.c_ulonglong * 16)()
for rep in range(2):
    g.preprocess_transition_probs()
    torch.cuda.synchronize()
    assert lib.n2v_debug_tab_stamps(buf, 1) == 0
v = list(buf)
names = ["weights", "sum", "normalise+stacks", "pairing", "emit", "hand-out/header"]
for label, off in (("tables in LDS (K <= 512)", 0), ("tables in global memory (K > 512)", 8)):
    tot = sum(v[off:off + 6])
    print("%s: %.3e slots, %.1f wave-cycles per slot" % (label, v[off + 6], tot / max(v[off + 6], 1)))
    for i, n in enumerate(names):
        print("   %-18s %5.1f %%  %.1f cycles/slot" % (n, 100.0 * v[off + i] / max(tot, 1), v[off + i] / max(v[off + 6], 1)))
