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
def rccl_world1():
    """A one-rank RCCL group standing in for the world: the collectives are real launches (capturable), the sums are the
    rank's own values."""
    import torch.distributed as dist
    from n2v_hip import dist as n2v_dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 100))
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    c = n2v_dist._Comm(False)
    c.world = WORLD
    return c


# eager = the Python loop (round 2); graph = one captured base interval replayed (round 3); *_rccl: through a one-rank
# RCCL group instead of the do-nothing communicator
VARIANTS = {"eager": ("eager loop, no wire", None, NoWire, False), "graph": ("graph replay, no wire", None, NoWire, True),
            "eager_rccl": ("eager loop, RCCL world-1 all-reduces", None, rccl_world1, False),
            "graph_rccl": ("graph replay, RCCL world-1 all-reduces", None, rccl_world1, True),
            "per_table": ("eager, one launch per table and step", PerTableOps(), NoWire, False)}
for name, ops, make_comm, graph in [VARIANTS[v] for v in os.environ.get("VARIANTS", "eager,graph,eager_rccl,graph_rccl").split(",")]:
    m = sgns.SgnsModel(cg.n_nodes, dim=128, window=10, negative=5, seed=1)
    m.build_vocab(counts=counts)
    comm = make_comm()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    mg = sgns.train(m, walks, lens, epochs=1, comm=comm, n_walks_global=n_global, shard_offset=0, merge="tsum", ops=ops,
                    timers=False, graph=graph)
    t_host = time.perf_counter() - t0
    b.record()
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    n_sub = sum(mg.n_merges[-1:])
    print("%s: world %d, %d walks local, merges per tier %s, graph replays %d | host %.2f s (%.0f us per sub-interval), "
          "wall %.2f s, stream %.2f s, pairs %.3e (%.3e /s), row sharing %s" % (
              name, WORLD, n_local, mg.n_merges, getattr(mg, "graph_replays", 0), t_host, t_host / max(n_sub, 1) * 1e6, t_all,
              a.elapsed_time(b) / 1e3, m.pairs_trained(), m.pairs_trained() / t_all, m.update_mode_name), flush=True)
    mg.release()
    del m, mg
# latency of a small all-reduce through torch.distributed + RCCL (one rank: launch + kernel, no wire)
import torch.distributed as dist
if dist.is_initialized():
    for nbytes in (4096, 65536, 1 << 20, 64 << 20):
        t = torch.zeros(nbytes // 2, dtype=torch.bfloat16, device="cuda")
        for _ in range(20):
            dist.all_reduce(t)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record()
        for _ in range(200):
            dist.all_reduce(t)
        b.record()
        host = (time.perf_counter() - t0) / 200
        torch.cuda.synchronize()
        print("world-1 RCCL all-reduce of %8d B: %.1f us stream time, %.1f us host time per call" % (
            nbytes, a.elapsed_time(b) / 200 * 1e3, host * 1e6), flush=True)
