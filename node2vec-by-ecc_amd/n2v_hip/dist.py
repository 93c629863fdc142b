"""One process per GPU (BASELINE config C4): walks shard by start vertex with
no collective, every rank trains a replica on its shard and the tables are merged over RCCL
(`torch.distributed`, backend "nccl" = RCCL on ROCm): by default pure sums at per-row cadences
(sgns.TieredSumMerger), optionally weighted sums at the cadence of sgns.auto_syncs (merge="hot").

Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
The reference's only parallelism on this path is the same split of start nodes into contiguous
chunks over pool workers (src/main_link.py:259-292); its workers share nothing while walking and
gensim's threads share one table while training — the merge emulates the latter across GPUs.
"""
import os

import torch

from . import sgns


class RankContext:
    """rank / world / device of this process and the communicator for the merges."""

    def __init__(self, backend="auto"):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        ndev = torch.cuda.device_count()
        if ndev == 0:
            raise RuntimeError("n2v_hip.dist: no GPU visible (there is no CPU fallback)")
        self.device = torch.device("cuda:%d" % (local_rank % ndev))
        torch.cuda.set_device(self.device)
        self.comm = None
        self.host_staged = False
        if self.world > 1:
            import torch.distributed as dist
            if backend == "auto":
                # fewer GPUs than ranks only happens when rehearsing on a one-GPU box: gloo, staged
                # through host memory (RCCL refuses two ranks on one device)
                backend = "nccl" if ndev >= self.world else "gloo"
            self.host_staged = backend == "gloo"
            if not dist.is_initialized():
                dist.init_process_group(backend, init_method="env://", rank=self.rank, world_size=self.world,
                                        device_id=None if self.host_staged else self.device)
            self.comm = _Comm(self.host_staged)

    def barrier(self):
        if self.world > 1:
            torch.distributed.barrier()

    def all_reduce_max(self, t):
        if self.world > 1:
            import torch.distributed as dist
            h = t.cpu() if self.host_staged else t
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            t.copy_(h)
        return t

    def close(self):
        if self.world > 1 and torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()


class _Comm(sgns._ProcessGroupComm):
    def __init__(self, host_staged):
        super().__init__()
        self.host_staged = host_staged
        self.graph_capturable = not host_staged        # the gloo rehearsal goes through host memory: eager only
        # over RCCL the replicas' changes travel as bfloat16 (half the bytes of every merge; AUC unchanged to 2e-4
        # on both probe graphs at 2 and 8 replicas); the gloo rehearsal path stays fp32 (gloo has no bf16 sum)
        self.wire_dtype = None if host_staged else torch.bfloat16
        self.wire_dtype_f64 = None if host_staged else torch.float32   # BiNE's fp64 tables: changes as fp32

    def all_reduce_sum(self, t):
        if self.host_staged:
            h = t.cpu()
            self.dist.all_reduce(h)
            t.copy_(h)
        else:
            self.dist.all_reduce(t)

    def all_reduce_async(self, t):
        """Start the all-reduce of `t`; the handle's wait() puts the sum into `t` in stream order.  Over RCCL the
        collective runs on the process group's stream under whatever is launched next; in the gloo rehearsal the
        host copy is reduced by gloo's own thread while the GPU trains."""
        if not self.host_staged:
            return self.dist.all_reduce(t, async_op=True)
        return _HostStagedWork(self.dist, t)


class _HostStagedWork:
    def __init__(self, dist, t):
        self.t = t
        self.h = t.cpu()       # waits for the kernels that produced `t` only: nothing else is queued yet
        self.work = dist.all_reduce(self.h, async_op=True)

    def wait(self):
        self.work.wait()
        self.t.copy_(self.h)


def shard_of(n_starts, ctx):
    """[pos_begin, pos_end) of this rank among the start positions (list(G.nodes()) order)."""
    return sgns.shard_bounds(n_starts, ctx.world, ctx.rank)


def sharded_walks(engine, ctx, num_walks, walk_length, seed, out=None):
    """This rank's walks: all `num_walks` rounds over its contiguous block of start positions,
    Philox uniforms keyed by the GLOBAL walk index — the union over ranks equals the single-GPU
    result row for row.  Returns (walks, lens, shard_offset) with shard_offset = number of
    sentences that precede this shard in shard-major order."""
    b, e = shard_of(int(engine.start_order.numel()), ctx)
    walks, lens = engine.walk(engine.start_order, num_walks, walk_length, rng="philox", seed=seed, pos_begin=b,
                              pos_count=e - b, out=out)
    return walks, lens, b * num_walks


def global_counts(walks, n_words, ctx):
    """Corpus word counts over all ranks (gensim's build_vocab scan), identical on every rank."""
    counts = torch.zeros(n_words, dtype=torch.int64, device=walks.device)
    for b in range(0, int(walks.shape[0]), 1 << 20):   # chunked: keeps the int64 temporaries small
        flat = walks[b:b + (1 << 20)].reshape(-1)
        counts += torch.bincount(flat[flat >= 0].long(), minlength=n_words)
    if ctx.world > 1:
        ctx.comm.all_reduce_sum(counts)
    return counts


def train_sharded(model, walks, lens, ctx, n_walks_global, shard_offset, epochs=1, syncs_per_epoch="auto",
                  merge="tsum", overlap=True, cold_delay=False):
    """sgns.train with this rank's communicator; afterwards every rank holds the merged tables.  Returns the
    merger (its timers: bench.py's merge_seconds / overlap_fraction)."""
    assert model.device == ctx.device, "replica on %s but this rank owns %s" % (model.device, ctx.device)
    return sgns.train(model, walks, lens, epochs=epochs, comm=ctx.comm, n_walks_global=n_walks_global,
                      shard_offset=shard_offset, syncs_per_epoch=syncs_per_epoch, merge=merge, overlap=overlap,
                      cold_delay=cold_delay)
