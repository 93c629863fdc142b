#!/usr/bin/env python3
"""Benchmark of the node2vec hot path on MI355X: 2nd-order walk + SGNS (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: `rounds` walks of length 80 from every
start vertex of this rank's shard (walk kernel), then one SGNS pass over those walks (SGNS
kernel; with N > 1 the replicas' tables are all-reduced over RCCL inside the pass).  Inputs
(CSR, alias tables, embedding tables, vocabulary statistics) are resident in HBM before the
timed region.  Scaling is STRONG by default (BASELINE config C4): the same rounds x N_nodes walks
at every N, start positions split over the ranks (src/main_link.py:261-264); `--scaling weak`
lets every rank walk rounds x N_nodes instead.

Prints ONE JSON line on rank 0 (see README / DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
WALK_BYTES_PER_STEP = 36     # table-driven walk, 32 B read + 4 B write
SGNS_BYTES_PER_PAIR_128 = 7168  # d=128, 1 positive + 5 negatives, fp32 read+write

CONFIGS = {
    # name: (graph key, p, q, description)
    "C2": ("C2", 1.0, 1.0, "C2: Erdos-Renyi 100k nodes / 1M edges, p=q=1, d=128"),
    "C3": ("C3", 0.25, 4.0, "C3: power-law (Barabasi-Albert) 1M nodes / 10M edges, p=0.25 q=4, d=128"),
    # BASELINE config 5 (BiNE path): its own pipeline and JSON line, see bench_bine()
    "C5": ("C5", None, None, "C5: bipartite 500k users + 500k items / 20M ratings, item popularity power-law, BiNE d=256"),
}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def cpu_baselines(cg, p, q, walks_sample, lens_sample, counts, dim, budget_s=12.0):
    """CPU restatement of the reference path timed on this host (rank 0 only, bounded sample).
    Primary: pure-Python restatement of src/node2vec.py (what the reference runs).  Extra: the
    oracle's C port (walk, 1 thread) and its SGNS restatement (1 thread and all cores)."""
    from oracle import c_oracle, sgns_oracle
    from oracle import n2v_oracle as orc
    out = {}
    G = orc.CsrBackedGraph(cg.labels, cg.row_ptr, cg.col, cg.w, cg.start_order, cg.directed)
    o = orc.Node2VecOracle(G, cg.directed, p, q)
    nodes = G.nodes[:2000]
    rs = np.random.RandomState(123)
    t0 = time.perf_counter()
    steps = 0
    done = 0
    for node in nodes:  # src/node2vec.py:97-111 order, one round
        w = o.node2vec_walk(80, node, rs.random_sample, on_the_fly=True)
        steps += len(w) - 1
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out["cpu_baseline"] = {
        "value": steps / dt, "unit": "walk-steps/s", "cores": 1, "kind": "port",
        "sample": "first %d start vertices x 1 round x L=80 of the same graph; pure-Python restatement of "
                  "src/node2vec.py (on-the-fly tables as src/settings.py:18, per-step sorted neighbours, two "
                  "MT19937 draws per step); %.1f s" % (done, dt)}
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    model = "?"
    try:
        model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        pass
    extra = {"host_cpus": os.cpu_count(), "affinity_cpus": affinity, "cpu_model": model}
    # the same restatement fanned out over processes by contiguous blocks of start nodes, as
    # src/main_link.py:259-292 does (spawned workers: this process owns a GPU context and must not fork)
    if cg.w is None:
        import subprocess
        import tempfile
        procs = max(1, min(16, affinity))
        with tempfile.TemporaryDirectory() as td:
            base = os.path.join(td, "g")
            for k, a in (("labels", cg.labels), ("row_ptr", cg.row_ptr), ("col", cg.col), ("start_order", cg.start_order)):
                np.save("%s_%s.npy" % (base, k), np.asarray(a))
            per = 400
            t0 = time.perf_counter()
            ps = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_workers", base, "1" if cg.directed else "0", str(p),
                                    str(q), str(r * per), str((r + 1) * per), "6.0", str(1000 + r)], cwd=ROOT,
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for r in range(procs)]
            res = []
            for pr in ps:
                try:
                    res.append(json.loads(pr.communicate(timeout=60)[0].strip().splitlines()[-1]))
                except Exception:
                    pr.kill()
            wall = time.perf_counter() - t0
        if res:
            busy = max(r[2] for r in res)
            extra["walk_python_processes"] = {
                "value": sum(r[0] for r in res) / busy, "unit": "walk-steps/s", "cores": len(res),
                "sample": "%d processes x up to %d start vertices each x 1 round (slowest worker %.1f s, %.1f s with "
                          "start-up)" % (len(res), per, busy, wall)}
    # C port of the walk (table rebuilt per step), 1 thread, 20000 starts
    co = c_oracle.CsrOracle(cg.row_ptr, cg.col, cg.w, p, q)
    t0 = time.perf_counter()
    ns = min(20000, cg.n_nodes)
    _, l, _ = co.walk(cg.start_order[:ns], 1, 80, mode="mt", seed=1, on_the_fly=True)
    dt = time.perf_counter() - t0
    extra["walk_c_port_on_the_fly"] = {"value": float((l - 1).sum()) / dt, "unit": "walk-steps/s", "cores": 1,
                                       "sample": "%d start vertices x 1 round" % ns}
    # the same C port on all cores: threads take contiguous blocks of start vertices (Philox keyed by walk index)
    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(64, affinity))
    blocks = [cg.start_order[k * ns // T:(k + 1) * ns // T] for k in range(T)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(T) as ex:
        outs = list(ex.map(lambda kb: co.walk(kb[1], 4, 80, mode="philox", seed=1, walk_index_base=kb[0] * 10**6,
                                              on_the_fly=True)[1], enumerate(blocks)))
    dt = time.perf_counter() - t0
    extra["walk_c_port_on_the_fly_allcores"] = {"value": float(sum((x - 1).sum() for x in outs)) / dt,
                                                "unit": "walk-steps/s", "cores": T,
                                                "sample": "%d start vertices x 4 rounds over %d threads" % (ns, T)}
    # SGNS restatement on a sample of the GPU-generated walks
    si, cum = sgns_oracle.vocab_tables(counts, 1e-3) if cg.n_nodes <= 200000 else (None, None)
    if si is None:
        from n2v_hip import sgns as _ps  # same statistics, vectorised (checked equal in tests/)
        si, cum = _ps.vocab_tables(counts, 1e-3)
    stride = 128 if dim <= 128 else 256
    # all cores: the threads this process may actually run on (affinity), and at least 4 jobs (gensim: <= 10 000 words
    # = 125 walks of 80) per thread so that every thread is busy for the whole measurement
    T_all = max(1, min(affinity, os.cpu_count() or 1, 64))
    for threads, nw in ((1, 2000), (T_all, max(16000, T_all * 4 * 125))):
        nw = min(nw, walks_sample.shape[0])
        syn0, syn1 = c_oracle.sgns_init(cg.n_nodes, dim, stride, 1)
        t0 = time.perf_counter()
        pairs = c_oracle.sgns_train(walks_sample[:nw], lens_sample[:nw], syn0, syn1, dim, 10, 5, si, cum,
                                    n_threads=threads)
        dt = time.perf_counter() - t0
        extra["sgns_c_port_%s" % ("1thread" if threads == 1 else "allcores")] = {
            "value": pairs / dt, "unit": "pair-updates/s", "cores": threads,
            "sample": "%d walks of the batch (%d jobs of <= 10 000 words, %.1f per thread)" % (nw, -(-nw // 125), -(-nw // 125) / threads)}
        del syn0, syn1
    out["cpu_baseline_extra"] = extra
    return out


def bench_bine(args):
    """`--config C5`: the BiNE path (src/bine_train.py) at BASELINE config 5.  A step = one training iteration:
    one pass over the 20M-rating list (skip-gram blocks of first-seen vertices + the KL update per rating), the
    replica merge when N > 1 (RCCL all-reduce of both tables) and the learning-rate step.  Strong scaling: the
    rating list is split among the ranks.  HITS, walks, pools and the occurrence index are built before the
    timed region (reported under "stages")."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bine_probe
    from n2v_hip import dist as n2v_dist
    ctx = n2v_dist.RankContext(backend=args.backend)
    pa = bine_probe.parser().parse_args([])
    pa.dim = args.dim if args.dim != 128 else 256
    pa.iters, pa.warmup_iters, pa.backend = args.steps, max(1, args.warmup), args.backend
    out = bine_probe.run(pa, ctx=ctx, emit=False)
    e, tr = out.pop("engine"), out["train"]
    if ctx.rank != 0:
        ctx.close()
        return
    K = tr["iterations"]
    launch_s = out["train_event_s"] / K
    bytes_launch = tr["rows_per_iteration"] * e.dim * 8
    traffic = None
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("C5_n1") if ctx.world == 1 else None
    except Exception:
        traffic = None
    result = {
        "metric": "BiNE rating-updates/s", "value": out["graph"]["ratings"] * K / out["train_s"],
        "unit": "rating-updates/s", "n_gpus": ctx.world, "steps": K, "warmup": pa.warmup_iters,
        "ms_per_step": out["train_s"] / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": CONFIGS["C5"][3], "ws": 5, "ns": 4, "maxT": pa.maxT, "minT": 1, "stop_probability": 0.15,
                   "dim": e.dim, "row_sharing": e.mode_used, **out["graph"], **out["walks"],
                   "sharding": "rating list split among ranks, replicas' changes summed over RCCL every iteration"
                   if ctx.world > 1 else "single GPU"},
        "stages_seconds": {k[:-2]: v for k, v in out.items() if k.endswith("_s") and not k.startswith("train")},
        "hits_iterations": out["hits_iterations"], "losses": tr["losses"][-K:],
        "negative_pools": {"method": "MinHash LSH forest (src/bine_lsh.py; datasketch 1.2.5 restated)", **out.get("neg_pools_lsh", {})},
        "roofline": {"kernel": "bine_train_kernel", "bound": "hbm", "achieved": bytes_launch / launch_s / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_launch / launch_s / 1e9 / HBM_PEAK_GBS,
                     "traffic": (traffic or {}).get("bine_train_kernel"),
                     "algorithmic_bytes_per_unit": bytes_launch / out["graph"]["ratings"], "unit_name": "rating",
                     "units_per_launch": out["graph"]["ratings"] // ctx.world, "launch_ms": launch_s * 1e3,
                     "accounting": "rows the kernel reads + writes (counted by the kernel) x dim x 8 B; the reference's "
                                   "own access pattern (every skip_gram call re-reads and re-writes all its rows) "
                                   "would move %.3g B per launch" % (tr["reference_pattern_rows_per_iteration"] * e.dim * 8)},
    }
    if not args.no_cpu_baseline and ctx.world == 1:
        result["cpu_baseline"] = bine_cpu_baseline(e)
    print(json.dumps(result), flush=True)
    ctx.close()


def bine_cpu_baseline(e, budget_s=15.0):
    """numpy restatement of src/bine_train.py:243-309,461-494 (oracle/bine_oracle.py) on every k-th rating of the
    same list with its real first-visit flags, against copies of the device's tables."""
    from n2v_hip import bine
    from oracle import bine_oracle as bo
    g = e.g
    host = {k: getattr(e, k).cpu().numpy() for k in ("occ_ptr", "occ_pos", "tokens", "tok_walk", "walk_off", "pool")}
    emb = e.emb[:, : e.dim].cpu().numpy().copy()
    ctx_t = e.ctx[:, : e.dim].cpu().numpy().copy()
    stride = max(1, g.n_ratings // 40000)
    idx = np.arange(0, g.n_ratings, stride)
    done, t0 = 0, time.perf_counter()
    for b in range(0, len(idx), 500):
        sel = idx[b:b + 500]
        bo.train(g.edge_u[sel], g.edge_v[sel], g.edge_w[sel], emb, ctx_t, host["occ_ptr"], host["occ_pos"], host["tokens"],
                 host["tok_walk"], host["walk_off"], host["pool"], 5, 4, 0.01, 0.01, 0.1, 0.01, 1,
                 bine.derive_seed(e.seed, bine.SEED_OCC), bine.derive_seed(e.seed, bine.SEED_NEG), first=g.first[sel])
        done += len(sel)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "rating-updates/s", "cores": 1, "kind": "port",
            "sample": "every %d-th rating of the same list (%d ratings, with their first-visit flags), one iteration; "
                      "numpy restatement of skip_gram / KL_divergence / the train loop; %.1f s" % (stride, done, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS))
    ap.add_argument("--rounds", type=int, default=10, help="walks per start vertex per step and per GPU (BASELINE: 10)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--syncs", default="auto", help="replica merges per SGNS pass when N > 1 (auto: staleness bound)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = BASELINE config C4 (the SAME rounds x N_nodes walks, start positions split "
                         "over the ranks, src/main_link.py:261-264); weak = every rank walks rounds x N_nodes")
    ap.add_argument("--merge", default="tsum", choices=["tsum", "hot", "delta", "avg"],
                    help="N > 1: replica merges — 'tsum' (default: pure sums at per-row cadences, inside the AUC band at every "
                         "size measured; many short launches) or 'hot' (per-row weights, 117 merges per pass at 8 GPUs: "
                         "several times faster, inside the band on small graphs only)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: wait for the cold rows' all-reduce at once instead of under the next interval (A/B)")
    ap.add_argument("--update-mode", default="auto", choices=["auto", "atomic", "agent", "plain"],
                    help="how racing wavefronts share embedding rows (DESIGN.md 4.3); auto = agent above 131072 rows")
    ap.add_argument("--allow-out-of-band", action="store_true",
                    help="permit modes measured outside the +-0.002 AUC band (--merge hot at this size, an explicit lossy "
                         "--update-mode on a short corpus); the default line never needs it")
    ap.add_argument("--merge-timers", action="store_true",
                    help="N > 1, --merge tsum: time the merges with events (runs the eager Python loop instead of replaying "
                         "the captured HIP graph of a base interval: slower, but merge_seconds is measured)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-shared-negatives", action="store_true", help="skip the extra opt-in SGNS variant pass")
    ap.add_argument("--no-reference-exact", action="store_true", help="skip the extra reference-exact walk pass")
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    if torch.cuda.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X; no GPU visible (there is no CPU fallback)")
    if args.config == "C5":
        return bench_bine(args)
    import node2vec
    from n2v_hip import dist as n2v_dist
    from n2v_hip import sgns, synth
    ctx = n2v_dist.RankContext(backend=args.backend)   # one process per GPU; RCCL unless rehearsing over gloo
    rank, dev, comm, barrier = ctx.rank, ctx.device, ctx.comm, ctx.barrier

    gkey, p, q, desc = CONFIGS[args.config]
    L, window, negative = 80, 10, 5
    t0 = time.perf_counter()
    cg, info = synth.make_config_graph(gkey)
    t_graph = time.perf_counter() - t0
    g = node2vec.Graph.from_csr(cg, p, q, device=dev, rng="philox", seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.preprocess_transition_probs()
    torch.cuda.synchronize()
    t_pre = time.perf_counter() - t0
    eng = g._engine
    t_pre_alloc = float(getattr(eng, "alloc_seconds", 0.0))    # host time of the table allocation (driver-side; see engine.py)
    # a second call: the tables come back from the allocator's cache, what remains is index plumbing + the kernels
    t0 = time.perf_counter()
    g.preprocess_transition_probs()
    torch.cuda.synchronize()
    t_pre_warm = time.perf_counter() - t0
    eng = g._engine
    if rank == 0:
        log("[bench] %s: graph %.1fs, preprocess %.2fs (%d alias slots, %.1f GB of %s slots)" % (
            args.config, t_graph, t_pre, eng.total_slots, eng.total_slots * (32 if eng.edge_fat is not None else 16) / 1e9,
            "fat 32-B" if eng.edge_fat is not None else "thin 16-B"))

    N = cg.n_nodes
    pos_begin, pos_end = sgns.shard_bounds(N, world, rank)
    pos_count = pos_end - pos_begin
    # strong (default, C4): the job is C3's rounds x N walks at every world size, a rank owns the start positions
    # [r*N/G, (r+1)*N/G) of every round; weak: every rank walks rounds x N (rounds x world in total)
    rounds_total = args.rounds * (world if args.scaling == "weak" else 1)
    n_local = pos_count * rounds_total
    n_global = N * rounds_total
    walks = torch.empty((n_local, L), dtype=torch.int32, device=dev)
    lens = torch.empty(n_local, dtype=torch.int32, device=dev)

    def walk_step(step_no):
        eng.walk(eng.start_order, rounds_total, L, rng="philox", seed=1000 + step_no, pos_begin=pos_begin,
                 pos_count=pos_count, out=(walks, lens))

    # vocabulary statistics (gensim build_vocab) from one batch of walks, identical on all ranks
    walk_step(-1)
    counts = n2v_dist.global_counts(walks, N, ctx)
    model = sgns.SgnsModel(N, dim=args.dim, window=window, negative=negative, seed=1, device=dev,
                           update_mode=args.update_mode, allow_out_of_band=args.allow_out_of_band)
    model.build_vocab(counts=counts)
    shard_offset = pos_begin * rounds_total
    syncs = (sgns.auto_syncs(n_global * L, N, world) if args.syncs == "auto" else int(args.syncs))

    merge_secs = {"merge": 0.0, "wait": 0.0, "n": 0}

    graph_mode = {"replays": 0}

    def sgns_step(step_no, timed=False):
        # tsum: the merges are timed only on request — timing them needs the eager loop; by default a base interval is
        # replayed from a captured HIP graph (n2v_hip/sgns.py:_train_tsum)
        timed = timed and (args.merge != "tsum" or args.merge_timers)
        mg = sgns.train(model, walks, lens, epochs=1, comm=comm, n_walks_global=n_global, shard_offset=shard_offset,
                        syncs_per_epoch=syncs if args.merge != "tsum" else "auto", merge=args.merge,
                        overlap=not args.no_overlap, timers=timed)
        if mg is not None:
            graph_mode["replays"] = int(getattr(mg, "graph_replays", 0))
            graph_mode["merges"] = mg.n_merges if isinstance(mg.n_merges, int) else sum(mg.n_merges)
        if timed and mg is not None:
            if mergers:
                mergers[-1].release()       # keep the timers, not 2.5 GB of snapshots per timed step
            mergers.append(mg)
        elif mg is not None:
            mg.release()

    mergers = []

    ev = lambda: torch.cuda.Event(enable_timing=True)
    for s in range(args.warmup):
        walk_step(s)
        sgns_step(s)
    torch.cuda.synchronize()
    model.pair_count.zero_()
    steps_done = torch.zeros(1, dtype=torch.int64, device=dev)
    marks = []
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        e0, e1, e2 = ev(), ev(), ev()
        e0.record()
        walk_step(args.warmup + s)
        e1.record()
        steps_done += (lens.long() - 1).clamp_min(0).sum()
        e1b = ev()
        e1b.record()
        sgns_step(args.warmup + s, timed=True)
        e2.record()
        marks.append((e0, e1, e1b, e2))
    torch.cuda.synchronize()
    barrier()
    t_total = time.perf_counter() - t0
    t_walk = sum(a.elapsed_time(b) for a, b, _, _ in marks) / 1e3
    t_sgns = sum(c.elapsed_time(d) for _, _, c, d in marks) / 1e3

    for mg in mergers:                    # compute-stream seconds inside the merge phases of the timed passes
        sec = mg.seconds()
        merge_secs["merge"] += sec["merge"]
        merge_secs["wait"] += sec["wait"]
        merge_secs["n"] += mg.n_merges if isinstance(mg.n_merges, int) else sum(mg.n_merges)   # tsum: all tiers' merges
    comm_probe = None
    if world > 1 and mergers:
        # what one merge's all-reduce costs when nothing runs beside it (for overlap_fraction): the wire buffer that
        # carries the bulk of the rows — the synchronous tier's unless most rows are in the delayed tier
        buf = mergers[-1].probe_buffer()
        comm.all_reduce_sum(buf)
        torch.cuda.synchronize()
        c0, c1 = ev(), ev()
        c0.record()
        for _ in range(3):
            comm.all_reduce_sum(buf)
        c1.record()
        torch.cuda.synchronize()
        comm_probe = c0.elapsed_time(c1) / 3e3
    merges_timed = bool(mergers)
    del mergers
    stats = ctx.all_reduce_max(torch.tensor([t_total, t_walk, t_sgns], dtype=torch.float64, device=dev))
    sums = torch.tensor([float(steps_done.item()), float(model.pairs_trained())], dtype=torch.float64, device=dev)
    if world > 1:
        comm.all_reduce_sum(sums)
    t_total, t_walk, t_sgns = [float(x) for x in stats.tolist()]
    steps_all, pairs_all = [float(x) for x in sums.tolist()]

    # opt-in variant, reported separately: negatives shared per centre word (one pass, N=1 only)
    shared = None
    if world == 1 and not args.no_shared_negatives:
        ms = sgns.SgnsModel(N, dim=args.dim, window=window, negative=negative, seed=1, device=dev, share_negatives=True,
                            allow_out_of_band=True)     # reported separately and labelled as not gensim's sampling
        ms.build_vocab(counts=counts)
        sgns.train(ms, walks[: max(1, n_local // 10)], lens[: max(1, n_local // 10)], epochs=1)   # warm-up
        ms.pair_count.zero_()
        s0, s1 = ev(), ev()
        s0.record()
        sgns.train(ms, walks, lens, epochs=1)
        s1.record()
        torch.cuda.synchronize()
        shared = {"metric": "SGNS pair-updates/s, negatives shared per centre word (opt-in; not gensim's sampling)",
                  "value": ms.pairs_trained() / (s0.elapsed_time(s1) / 1e3), "unit": "pair-updates/s",
                  "seconds": s0.elapsed_time(s1) / 1e3}
        del ms
    # the reference-exact mode, reported separately (N=1): numpy's global MT19937 stream regenerated
    # on the device, end to end (uniform generation + walk), same walks as the reference for this seed
    exact = None
    if world == 1 and not args.no_reference_exact:
        g.rng = "numpy"
        np.random.seed(123)
        g.simulate_walks(args.rounds, L)             # warm-up at full size (jump polynomials, the allocator's 12.6 GB buffer)
        times = []
        for _ in range(3):
            np.random.seed(123)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            corpus = g.simulate_walks(args.rounds, L)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]     # median of three: a run that has to re-allocate its 12.6 GB buffer on a box whose
        # driver wipes freed memory first (65-82 ms/GiB) is 10x slower than the others; all three are reported
        exact = {"metric": "walk-steps/s, reference-exact mode (np.random.seed(123) stream, generated on the GPU)",
                 "value": float((corpus.lens.long() - 1).sum().item()) / dt, "unit": "walk-steps/s", "seconds": dt,
                 "seconds_each": times, "seconds_is": "median of seconds_each", "uniform_layout": "tiled (64 walks x step-major), one chunk"}
        del corpus
        g.rng = "philox"
    # the reference's memory-saving mode (ON_THE_FLY = True, src/settings.py:18, is main_link's default): no stored edge
    # tables, every step rebuilds — or proves it does not need — its alias table (one round of the same walks, N=1)
    otf = None
    if world == 1 and not args.no_reference_exact:
        eng.walk_on_the_fly(eng.start_order, 1, L, rng="philox", seed=3)      # warm-up (scratch rows)
        o0, o1 = ev(), ev()
        o0.record()
        _, ol = eng.walk_on_the_fly(eng.start_order, 1, L, rng="philox", seed=4)
        o1.record()
        torch.cuda.synchronize()
        otf = {"metric": "walk-steps/s, on-the-fly tables (src/node2vec.py:97-111: no stored edge tables)",
               "value": float((ol.long() - 1).sum().item()) / (o0.elapsed_time(o1) / 1e3), "unit": "walk-steps/s",
               "seconds": o0.elapsed_time(o1) / 1e3, "rounds": 1}
        del ol
    if rank != 0:
        ctx.close()
        return
    K = args.steps
    walk_rate = steps_all / t_walk
    pair_rate = pairs_all / t_sgns
    stride_scale = model.stride / 128.0
    # per-launch figures of THIS rank (rank 0): algorithmic bytes / mean launch duration
    walk_launch_s = sum(a.elapsed_time(b) for a, b, _, _ in marks) / 1e3 / K
    sgns_launch_s = sum(c.elapsed_time(d) for _, _, c, d in marks) / 1e3 / K
    if world > 1:
        # the kernel's share of the pass: the stream time between the marks minus what the merges took of it
        sgns_launch_s = max(sgns_launch_s - merge_secs["merge"] / K, 1e-9)
    walk_bytes_launch = float(steps_done.item()) / K * WALK_BYTES_PER_STEP
    sgns_bytes_launch = float(model.pairs_trained()) / K * SGNS_BYTES_PER_PAIR_128 * stride_scale
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = tj.get("%s_n1" % args.config) if world == 1 else None
            if traffic:
                traffic_source = ("profiles/traffic.json (%s): rocprofv3 PMC passes of tools/run_c3_profile.sh, NOT measured "
                                  "by this run" % tj.get("_round", "r01"))
        except Exception:
            traffic = None
    wire_desc = ("gloo through host memory, fp32 wire — a rehearsal, not RCCL" if getattr(ctx, "host_staged", False)
                 else "RCCL over xGMI, %s wire" % str(getattr(ctx.comm, "wire_dtype", None) or torch.float32).replace("torch.", ""))
    result = {
        "metric": "walk-steps/s",
        "value": walk_rate,
        "unit": "walk-steps/s",
        "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": t_total / K * 1e3,
        "higher_is_better": True, "scaling": args.scaling if world > 1 else "strong", "vs_baseline": None,
        "dtype": "f64 alias draw / int32 ids (walk); f32 (SGNS)",
        "data": "synthetic",
        "config": {"workload": desc, "walk_length": L, "rounds_per_step": rounds_total,
                   "walks_per_step_global": n_global, "window": window, "negative": negative, "dim": args.dim,
                   "rng": "philox4x32-10 in-kernel (walk rule bit-identical to the reference under given uniforms)",
                   "sharding": "single GPU" if world == 1 else (
                       ("start-vertex shards, %d 'hot'-weighted synchronous merges per SGNS pass, pipelined over row "
                        "ranges (%s)" % (syncs, wire_desc)) if args.merge == "hot" else
                       "start-vertex shards, merge=tsum: pure-sum merges at per-row cadences (%d tiers x%d, base "
                       "staleness budget %d), one all-reduce (%s) per merge" % (
                           sgns.TSUM_TIERS, sgns.TSUM_RATIO, sgns.TSUM_STALENESS_BUDGET, wire_desc) if args.merge == "tsum" else
                       "start-vertex shards, merge=%s" % args.merge), **info},
        "sgns": {"metric": "SGNS pair-updates/s", "value": pair_rate, "unit": "pair-updates/s",
                 "row_sharing": model.update_mode_name,
                 "pairs_per_step_global": pairs_all / K, "seconds_per_step": t_sgns / K},
        "walk": {"steps_per_step_global": steps_all / K, "seconds_per_step": t_walk / K,
                 "table_layout": "fat (32-B slots)" if eng.edge_fat is not None else "thin (16-B slots + records)"},
        "walk_reference_exact": exact,
        "walk_on_the_fly": otf,
        "sgns_shared_negatives": shared,
        # first call in this process, of which the table allocation (hipMalloc, host-blocking when the driver hands out
        # memory some allocation has just released), the rest (kernels + index plumbing), and a second call on the
        # same graph (tables reused from the allocator's cache)
        "preprocess_seconds": t_pre, "preprocess_alloc_seconds": t_pre_alloc,
        "preprocess_kernel_seconds": t_pre - t_pre_alloc, "preprocess_seconds_second_call": t_pre_warm,
        "alias_slots": eng.total_slots,
        # N > 1 (rank 0's timers): seconds per step the compute stream spent in the merge phases, of which waiting
        # for the cold rows' all-reduce; overlap_fraction = share of that all-reduce's stand-alone cost that was
        # hidden under training
        "merge_seconds": merge_secs["merge"] / K if world > 1 and merges_timed else None,       # None: not timed (graph replay)
        "merge_wait_seconds": merge_secs["wait"] / K if world > 1 and merges_timed else None,
        "merges_per_step": (merge_secs["n"] / K if merge_secs["n"] else graph_mode.get("merges")) if world > 1 else None,
        # tsum without --merge-timers: one base interval captured as a HIP graph and replayed this many times per pass
        "merge_graph_replays_per_step": graph_mode["replays"] if world > 1 else None,
        "allreduce_seconds_standalone": comm_probe,
        # tsum merges are synchronous (pack -> all-reduce -> apply between two training launches): nothing is hidden
        "overlap_fraction": (None if not comm_probe else 0.0 if args.merge == "tsum" else
                             max(0.0, 1.0 - merge_secs["wait"] / max(comm_probe * merge_secs["n"], 1e-12))),
        # dominant kernel by time: sgns_kernel
        "roofline": {"kernel": "sgns_kernel", "bound": "hbm", "achieved": sgns_bytes_launch / sgns_launch_s / 1e9,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": sgns_bytes_launch / sgns_launch_s / 1e9 / HBM_PEAK_GBS,
                     "traffic": (traffic or {}).get("sgns_kernel"), "traffic_source": traffic_source,
                     "algorithmic_bytes_per_unit": SGNS_BYTES_PER_PAIR_128 * stride_scale, "unit_name": "pair",
                     "launch_ms": sgns_launch_s * 1e3},
        "roofline_walk": {"kernel": "walk_fat2_kernel" if eng.edge_fat is not None else "walk_kernel", "bound": "hbm", "achieved": walk_bytes_launch / walk_launch_s / 1e9,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": walk_bytes_launch / walk_launch_s / 1e9 / HBM_PEAK_GBS,
                          "traffic": (traffic or {}).get("walk_kernel"), "traffic_source": traffic_source,
                          "algorithmic_bytes_per_unit": WALK_BYTES_PER_STEP, "unit_name": "walk step",
                          "launch_ms": walk_launch_s * 1e3,
                          # one 64-B request per step; the fabric serves ~5.05e10 random requests/s from a table of this
                          # size (tools/lab/gather_lab2.hip, profiles/r02/logs/lab2_gather2.jsonl): the ceiling of a
                          # one-gather-per-step walk is 5.05e10 x 36 B / 8 TB/s = 22.7 % by this accounting
                          "request_ceiling_steps_per_s": 5.05e10,
                          "frac_of_request_ceiling": float(steps_done.item()) / K / walk_launch_s / 5.05e10},
    }
    if not args.no_cpu_baseline and world == 1:      # the CPU baseline is timed at N = 1 only
        t0 = time.perf_counter()
        ns = min(32000, n_local)
        result.update(cpu_baselines(cg, p, q, walks[:ns].cpu().numpy(), lens[:ns].cpu().numpy(),
                                    model.counts, args.dim))
        log("[bench] cpu baselines took %.1fs" % (time.perf_counter() - t0))
    print(json.dumps(result), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
