"""Scratch probe: SGNS pass time on C3 walks as a function of the launch grid (max_blocks) and of the number of
walks per launch (what a multi-GPU merge interval looks like)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch
import node2vec
from n2v_hip import sgns, synth

cg, info = synth.make_config_graph("C3")
g = node2vec.Graph.from_csr(cg, 0.25, 4.0, rng="philox", seed=1)
g.preprocess_transition_probs()
corpus = g.simulate_walks(2, 80)
m = sgns.SgnsModel(cg.n_nodes, dim=128, window=10, negative=5, seed=1, update_mode=os.environ.get("MODE", "agent"))   # bench's C3 mode
m.build_vocab(corpus.walks)
W = corpus.walks.shape[0]


def run(n_walks, mb, reps=1):
    m.pair_count.zero_()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for r in range(reps):
        b = (r * n_walks) % (W - n_walks + 1)
        m.train_pass(corpus.walks[b:b + n_walks], corpus.lens[b:b + n_walks], 0, W, b, max_blocks=mb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return dt, m.pairs_trained() / dt


run(W, 0)
for mb in [int(x) for x in os.environ.get("GRIDS", "1024,2048,3072,4096,6144").split(",")]:
    dt, rate = run(W, mb)
    print("full pass %7d walks  max_blocks %5d: %.3f s  %.3e pairs/s" % (W, mb, dt, rate), flush=True)
for n in ([] if os.environ.get("FULL_ONLY") else [5356, 10700, 21400]):
    for mb in (2048, 3072, 4096):
        dt, rate = run(n, mb, reps=40)
        print("interval  %7d walks  max_blocks %5d: %.2f ms per launch  %.3e pairs/s" % (n, mb, dt / 40 * 1e3, rate), flush=True)
