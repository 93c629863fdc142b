"""preprocess_transition_probs on C3, table-kernel time (N2V_TIMING phases), for the library N2V_HIP_LIB points at."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch
import node2vec
from n2v_hip import synth, _lib
cg, info = synth.make_config_graph("C3")
os.environ["N2V_TIMING"] = "1"
g = node2vec.Graph.from_csr(cg, 0.25, 4.0, device="cuda:0", rng="philox", seed=1)
for rep in range(3):
    g.preprocess_transition_probs()
    torch.cuda.synchronize()
print(os.path.basename(_lib.SO_PATH), {k: round(v, 4) for k, v in g._engine.timings.items()}, flush=True)
c = g.simulate_walks(2, 80)
print("  walk checksum", int(c.walks.long().sum().item()))
