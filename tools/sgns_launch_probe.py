"""SGNS pair rate as a function of the launch size (walks per n2v_sgns_train call) on C3's walks: what the short
launches between two hub-tier merges of merge="tsum" cost (one wavefront trains one walk at a time)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch

import node2vec
from n2v_hip import sgns, synth

cg, info = synth.make_config_graph("C3")
g = node2vec.Graph.from_csr(cg, 0.25, 4.0, device="cuda:0", rng="philox", seed=1)
g.preprocess_transition_probs()
corpus = g.simulate_walks(2, 80)
m = sgns.SgnsModel(cg.n_nodes, dim=128, window=10, negative=5, seed=1, update_mode="agent")
m.build_vocab(corpus.walks)
W = corpus.walks.shape[0]
SIZES = [int(x) for x in os.environ.get("SIZES", "83,334,1335,2298,5342,21368,200000").split(",")]
print("N2V_SGNS_PREDRAW =", os.environ.get("N2V_SGNS_PREDRAW"), flush=True)
for size, splits in [(sz, sp) for sz in SIZES for sp in (1, "auto")]:
    n_launch = max(4, min(200, 400000 // size))
    m.pair_count.zero_()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n_launch):
        lo = (i * size) % (W - size)
        m.train_pass(corpus.walks[lo:lo + size], corpus.lens[lo:lo + size], sentences_base=lo, sentences_total=W, walk_id_base=lo,
                     splits=splits)
    b.record()
    torch.cuda.synchronize()
    dt = a.elapsed_time(b) / 1e3
    print("launch of %6d walks, walk_splits %-4s: %.3e pairs/s (%d launches, %.1f us per launch)" % (
        size, splits, m.pairs_trained() / dt, n_launch, dt / n_launch * 1e6), flush=True)
