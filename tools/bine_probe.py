#!/usr/bin/env python3
"""Timing of the BiNE path at BASELINE config 5's shape (or a scaled-down one) on one MI355X.

    python tools/bine_probe.py [--users 500000 --items 500000 --ratings 20000000 --dim 256 --iters 5]
Prints one JSON object: per-stage seconds and the training pass's achieved algorithmic bandwidth
(rows read + written, counted by the kernel, x dim x 8 B / pass time; peak 8 TB/s)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))

import numpy as np
import torch


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=500_000)
    ap.add_argument("--items", type=int, default=500_000)
    ap.add_argument("--ratings", type=int, default=20_000_000)
    ap.add_argument("--dim", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--warmup-iters", type=int, default=1)
    ap.add_argument("--maxT", type=int, default=32)
    ap.add_argument("--pool", type=int, default=200)
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--mode", default="parallel", choices=["parallel", "atomic", "store"])
    ap.add_argument("--backend", default="auto", choices=["auto", "nccl", "gloo"])
    ap.add_argument("--overlap-merge", action="store_true",
                    help="N > 1: the all-reduce of a pass's changes runs under the next pass (OverlappedReplicaMerge)")
    ap.add_argument("--split", action="store_true", help="also time a pass with the skip-gram blocks disabled (KL only)")
    return ap


def run(args, ctx=None, emit=True):
    """The whole BiNE pipeline once, timed per stage; returns the result dict (rank 0 prints it when emit)."""
    from n2v_hip import bine, synth
    from n2v_hip import dist as n2v_dist
    own_ctx = ctx is None
    if own_ctx:
        ctx = n2v_dist.RankContext(backend=args.backend)   # one process per GPU under torch.distributed.run
    dev = str(ctx.device)

    out = {"config": vars(args), "n_gpus": ctx.world}
    sync = torch.cuda.synchronize

    def timed(name, fn):
        sync()
        ctx.barrier()
        t0 = time.perf_counter()
        r = fn()
        sync()
        ctx.barrier()
        out[name + "_s"] = time.perf_counter() - t0
        if ctx.rank == 0:
            print("[bine] %-22s %.3f s" % (name, out[name + "_s"]), file=sys.stderr, flush=True)
        return r

    t0 = time.perf_counter()
    u, i, r = synth.bipartite_powerlaw_ratings(args.users, args.items, args.ratings)
    g = bine.BipartiteGraph(u, i, r)
    out["host_graph_s"] = time.perf_counter() - t0
    deg = np.diff(g.row_ptr)
    out["graph"] = {"n_u": g.n_u, "n_v": g.n_v, "ratings": g.n_ratings, "nnz": int(g.col.shape[0]),
                    "max_deg_u": int(deg[: g.n_u].max()), "max_deg_v": int(deg[g.n_u:].max())}
    if ctx.rank == 0:
        print("[bine] graph", out["graph"], "%.1f s" % out["host_graph_s"], file=sys.stderr, flush=True)
    e = timed("upload", lambda: bine.BineEngine(g, device=dev, seed=42))

    def run(**kw):
        if ctx.world > 1:
            return e.train_sharded(ctx.comm, ctx.rank, ctx.world, overlap=getattr(args, "overlap_merge", False), **kw)
        return e.train(**kw)

    timed("hits", e.calculate_centrality)
    out["hits_iterations"] = e.hits_iterations
    timed("walks", lambda: e.generate_walks(0.15, args.maxT, 1))
    out["walks"] = {"n_walks_u": e.n_walks[0], "n_walks_v": e.n_walks[1], "tokens": int(e.tokens.shape[0])}
    timed("neg_pools", lambda: e.build_negative_pools(args.pool))
    if getattr(e, "lsh", None):
        out["neg_pools_lsh"] = {"seconds": e.lsh["seconds"], "clusters_u": e.lsh.get("u", {}).get("clusters"),
                                "clusters_v": e.lsh.get("v", {}).get("clusters"),
                                "mean_query_size_u": float(e.lsh["u"]["sim_n"].float().mean().item()) if "u" in e.lsh else None,
                                "mean_query_size_v": float(e.lsh["v"]["sim_n"].float().mean().item()) if "v" in e.lsh else None}
        if ctx.rank == 0:
            print("[bine] lsh", out["neg_pools_lsh"], file=sys.stderr, flush=True)
    timed("occurrences", e.build_occurrences)
    timed("init", lambda: e.init_embeddings(args.dim))
    timed("train_warmup", lambda: run(max_iter=max(1, args.warmup_iters), max_blocks=args.max_blocks, mode=args.mode))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    losses = timed("train", lambda: run(max_iter=args.iters, max_blocks=args.max_blocks, mode=args.mode))
    ev1.record()
    sync()
    out["train_event_s"] = ev0.elapsed_time(ev1) / 1e3
    rows, rows_ref = float(e.state[4].item()), float(e.state[5].item())
    per = out["train_s"] / len(losses)
    out["train"] = {"iterations": len(losses), "seconds_per_iteration": per, "losses": losses, "lam": e.lam, "mode": e.mode_used,
                    "rows_per_iteration": rows / len(losses),
                    "algorithmic_GBps": rows / len(losses) * e.dim * 8 / per / 1e9,
                    "frac_of_8TBps": rows / len(losses) * e.dim * 8 / per / 8e12,
                    "reference_pattern_rows_per_iteration": rows_ref / len(losses),
                    "reference_pattern_GBps": rows_ref / len(losses) * e.dim * 8 / per / 1e9,
                    "ratings_per_s": g.n_ratings / per}
    if args.split and ctx.world == 1:
        first = e.first.clone()
        e.first.zero_()
        rows0 = float(e.state[4].item())
        timed("train_kl_only", lambda: e.train(max_iter=args.iters, max_blocks=args.max_blocks, first_iteration=100))
        r1 = float(e.state[4].item()) - rows0
        out["kl_only"] = {"seconds_per_iteration": out["train_kl_only_s"] / args.iters, "rows_per_iteration": r1 / args.iters,
                          "algorithmic_GBps": r1 / args.iters * e.dim * 8 / (out["train_kl_only_s"] / args.iters) / 1e9}
        e.first.copy_(first)
        for md in ("atomic", "store"):
            timed("train_mode_%s" % md, lambda: e.train(max_iter=2, mode=md, first_iteration=200))
    out["engine"] = e
    if emit and ctx.rank == 0:
        print(json.dumps({k: v for k, v in out.items() if k != "engine"}))
    if own_ctx:
        ctx.close()
    return out


def main():
    run(parser().parse_args())


if __name__ == "__main__":
    main()
