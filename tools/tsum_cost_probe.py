"""What one rank of an 8-GPU C4 run does per SGNS pass under merge="tsum", minus the wire: the rank-0 shard of C3's
walks (1.25e6 of 1e7 walks) trained through n2v_hip.sgns.train with a stand-in communicator of world size 8 whose
all-reduce does nothing.  Reports host wall time (launch-bound or not), stream time, and the merge kernels' share —
with all tables of a merge in one launch (the shipped path) and with one launch per table and step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch

import node2vec
from n2v_hip import sgns, synth

WORLD = int(os.environ.get("WORLD", "8"))


class NoWire:
    wire_dtype = torch.bfloat16
    world, rank = WORLD, 0

    def all_reduce_async(self, t):
        return None

    def all_reduce_sum(self, t):
        pass


class PerTableOps:
    """HipMergeOps without the fused tsum entry points: TieredSumMerger falls back to one launch per table."""

    def __init__(self):
        h = sgns.HipMergeOps()
        self.pack_rows, self.hot_apply = h.pack_rows, h.hot_apply


cg, info = synth.make_config_graph("C3")
g = node2vec.Graph.from_csr(cg, 0.25, 4.0, device="cuda:0", rng="philox", seed=1)
g.preprocess_transition_probs()
corpus = g.simulate_walks(2, 80)
n_global = cg.n_nodes * 10
n_local = n_global // WORLD
assert n_local <= corpus.walks.shape[0]
walks, lens = corpus.walks[:n_local].contiguous(), corpus.lens[:n_local].contiguous()
counts = torch.bincount(corpus.walks.reshape(-1)[corpus.walks.reshape(-1) >= 0].long(), minlength=cg.n_nodes) * 5
VARIANTS = {"fused": ("one launch per merge step", None), "per_table": ("one launch per table and step", PerTableOps())}
for name, ops in [VARIANTS[v] for v in os.environ.get("VARIANTS", "fused,per_table").split(",")]:
    m = sgns.SgnsModel(cg.n_nodes, dim=128, window=10, negative=5, seed=1)
    m.build_vocab(counts=counts)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    prof = None
    if os.environ.get("HOST_PROFILE"):
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    a.record()
    mg = sgns.train(m, walks, lens, epochs=1, comm=NoWire(), n_walks_global=n_global, shard_offset=0, merge="tsum", ops=ops, timers=bool(os.environ.get("TIMERS", "1") == "1"))
    t_host = time.perf_counter() - t0
    if prof is not None:
        prof.disable()
        import pstats
        pstats.Stats(prof).sort_stats("tottime").print_stats(18)
    b.record()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    sec = mg.seconds()
    n_sub = sum(mg.n_merges[-1:])
    print("%s: world %d, %d walks local, merges per tier %s | host issue %.2f s (%.0f us per sub-interval), "
          "wall %.2f s, stream %.2f s, merge kernels %.2f s, pairs %.3e (%.3e /s)" % (
              name, WORLD, n_local, mg.n_merges, t_host, t_host / max(n_sub, 1) * 1e6, t_all, a.elapsed_time(b) / 1e3,
              sec["merge"], m.pairs_trained(), m.pairs_trained() / t_all), flush=True)
    mg.release()
    del m, mg
