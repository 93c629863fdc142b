"""Lab: ASYNCHRONOUS pure-sum merges — what hiding the exchange completely would look like.  G replicas train their
whole shard in ONE launch each, concurrently (few workgroups per replica so that all of them run side by side and the
pass lasts long enough), while a merge loop on another stream keeps exchanging: d_r = x_r - known_r (bf16), S = sum_r d_r,
x_r += S - d_r by ATOMIC adds into the live table (the training kernel runs in its atomic mode, so nothing is lost),
known_r += S.  Hot rows (SumTierPlan tier >= 1) in every iteration, all rows every `full_every` iterations.  Sums commute,
so a replica sees the others' changes as just more Hogwild traffic with a staleness of one loop period.
Usage: python tests/probes/async_lab.py <hub|pp> ; env GS, BLOCKS (workgroups per replica), FULL_EVERY."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "node2vec-by-ecc_amd"), os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, p)
import numpy as np
import torch

import node2vec
from n2v_hip import linkpred, sgns
import replica_auc_probe as rap

kind = sys.argv[1] if len(sys.argv) > 1 else "hub"
seq = {"pp": 0.89607, "hub": 0.86678}.get(kind)
g, te, neg = rap.setup(kind)
Gr = node2vec.Graph.from_csr(g, 1.0, 1.0, rng="philox", seed=1)
Gr.preprocess_transition_probs()
rounds, L = 10, 80
corpus = Gr.simulate_walks(rounds, L)
te_d = np.stack([g.dense_of(te[:, 0]), g.dense_of(te[:, 1])], 1)
neg_d = np.stack([g.dense_of(neg[:, 0]), g.dense_of(neg[:, 1])], 1)
n = g.n_nodes
counts = torch.bincount(corpus.walks.reshape(-1).long(), minlength=n)
n_global = corpus.walks.shape[0]
dev = corpus.walks.device

for G in [int(x) for x in os.environ.get("GS", "8").split(",")]:
    for blocks in [int(x) for x in os.environ.get("BLOCKS", "4").split(",")]:
        for full_every in [int(x) for x in os.environ.get("FULL_EVERY", "8").split(",")]:
            models, shards = [], []
            for r in range(G):
                m = sgns.SgnsModel(n, dim=128, window=10, negative=5, seed=1, update_mode="atomic")
                m.build_vocab(counts=counts)
                models.append(m)
                b, e = sgns.shard_bounds(n, G, r)
                idx = (torch.arange(rounds, device=dev)[:, None] * n + torch.arange(b, e, device=dev)[None, :]).reshape(-1)
                shards.append((corpus.walks[idx].contiguous(), corpus.lens[idx].contiguous(), b * rounds))
            plan = sgns.SumTierPlan(counts.cpu().numpy(), n_global * L / 234.0, G, 10, 5, dev)
            names = ("syn0", "syn1neg")
            hot = [plan.rows_ge[ti][1] for ti in range(2)]
            allrows = [plan.rows_ge[ti][0] for ti in range(2)]
            known = [[getattr(m, nm).clone() for nm in names] for m in models]
            streams = [torch.cuda.Stream() for _ in range(G)]
            sm = torch.cuda.Stream()
            torch.cuda.synchronize()
            t0 = time.time()
            done = []
            for r, m in enumerate(models):
                w, l, off = shards[r]
                with torch.cuda.stream(streams[r]):
                    # one launch for the whole shard; sentences_step = G keeps the learning-rate schedule global
                    m.train_pass(w, l, sentences_base=0, sentences_step=G, sentences_total=n_global, walk_id_base=off,
                                 max_blocks=blocks)
                    ev = torch.cuda.Event()
                    ev.record()
                    done.append(ev)
            iters, fulls = 0, 0
            with torch.cuda.stream(sm):
                while not all(ev.query() for ev in done):
                    iters += 1
                    full = iters % full_every == 0
                    fulls += int(full)
                    for ti, nm in enumerate(names):
                        rows = allrows[ti] if full else hot[ti]
                        if rows.numel() == 0:
                            continue
                        ds = [(getattr(m, nm)[rows] - known[r][ti][rows]).bfloat16().float() for r, m in enumerate(models)]
                        S = torch.stack(ds).sum(0)
                        for r, m in enumerate(models):
                            getattr(m, nm).index_add_(0, rows, S - ds[r])        # atomic adds into the live table
                            known[r][ti].index_add_(0, rows, S)
                    sm.synchronize()          # one exchange at a time, like one collective in flight
            torch.cuda.synchronize()
            t_train = time.time() - t0
            # final exact exchange of everything
            for ti, nm in enumerate(names):
                ds = [getattr(m, nm) - known[r][ti] for r, m in enumerate(models)]
                S = torch.stack(ds).sum(0)
                for r, m in enumerate(models):
                    getattr(m, nm).copy_(known[r][ti] + S)
            spread = max(float((getattr(models[r], "syn0") - models[0].syn0).abs().max()) for r in range(1, G))
            auc = linkpred.get_roc_score(models[0].vectors(), te_d, neg_d)[0]
            print("[%s] G=%d blocks/replica=%d: pass %.2fs, %d exchanges (%d of all rows; hot rows syn0 %d syn1neg %d): AUC %.5f (%+.5f)"
                  "  replica spread %.1e" % (kind, G, blocks, t_train, iters, fulls, hot[0].numel(), hot[1].numel(), auc, auc - seq, spread),
                  flush=True)
