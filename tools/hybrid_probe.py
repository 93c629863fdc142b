"""Tables under a memory budget on C3: walk rate of the hybrid kernel (stored tables for the low-degree destinations,
per-step rebuild for the rest) at several budgets, next to the pure on-the-fly walk and the full tables."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch
import node2vec
from n2v_hip import synth

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cg, info = synth.make_config_graph(name)
g = node2vec.Graph.from_csr(cg, 0.25, 4.0, device="cuda:0", rng="philox", seed=1)
eng = g._graph_engine()
full = eng.total_edge_slots * 32
n_walks = 200000
starts = eng.start_order[:n_walks].contiguous()


def rate(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.perf_counter()
        w, l = fn()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return float((l.long() - 1).sum().item()) / best, int(w.long().sum().item())

r, chk = rate(lambda: eng.walk_on_the_fly(starts, 1, 80, rng="philox", seed=7))
print("%s pure on-the-fly: %.3e steps/s (checksum %d)" % (name, r, chk), flush=True)
ref = chk
for frac in (1 / 16, 1 / 8, 1 / 4, 1 / 3, 1 / 2, 3 / 4, 1.0):
    t = time.perf_counter()
    g.preprocess_transition_probs(budget_bytes=int(full * frac))
    torch.cuda.synchronize()
    tp = time.perf_counter() - t
    e = g._engine
    r, chk = rate(lambda: e.walk(starts, 1, 80, rng="philox", seed=7))
    stored_share = e.total_slots * 32 / full
    steps_tab = float((e.deg[e.col.long()] <= (e.stored_degree_cut if e.partial else 1 << 30)).float().mean().item())
    print("%s budget %.3f of the full tables (%.1f GB): degree cut %s, %.1f%% of the slots, %.1f%% of the entries stored; "
          "preprocess %.2f s; walk %.3e steps/s; identical %s" % (name, frac, full * frac / 1e9, e.stored_degree_cut, 100 * stored_share,
          100 * steps_tab, tp, r, chk == ref), flush=True)
