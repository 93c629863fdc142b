"""Scratch probe: link-prediction AUC when G replicas train on start-vertex shards and are
merged (delta / avg) `syncs` times per pass — the multi-GPU scheme of n2v_hip/sgns.py:train —
simulated on ONE GPU by training the replicas' chunks one after another."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import node2vec
from n2v_hip import linkpred, sgns
from test_gpu_sgns import _auc_setup


def simulate(G, corpus, n_nodes, rounds, merge, syncs, mode="atomic"):
    n = corpus.walks.shape[0] // rounds
    models = [sgns.SgnsModel(n_nodes, dim=128, window=10, negative=5, seed=1, update_mode=mode) for _ in range(G)]
    counts = torch.bincount(corpus.walks.reshape(-1).long(), minlength=n_nodes)
    for m in models:
        m.build_vocab(counts=counts)
    shards = []
    for r in range(G):
        b, e = sgns.shard_bounds(n, G, r)
        idx = (torch.arange(rounds, device=corpus.walks.device)[:, None] * n +
               torch.arange(b, e, device=corpus.walks.device)[None, :]).reshape(-1)
        shards.append((corpus.walks[idx].contiguous(), corpus.lens[idx].contiguous(), b * rounds))
    n_global = corpus.walks.shape[0]
    bases = [models[0].syn0.clone(), models[0].syn1neg.clone()]
    for c in range(syncs):
        for r, m in enumerate(models):
            w, l, off = shards[r]
            b, e = sgns.shard_bounds(w.shape[0], syncs, c)
            if e > b:
                m.train_pass(w[b:e], l[b:e], sentences_base=b * G, sentences_step=G, sentences_total=n_global,
                             walk_id_base=off + b)
        for ti, name in enumerate(("syn0", "syn1neg")):
            stack = torch.stack([getattr(m, name) for m in models])
            if merge.startswith("hot"):
                # per-row interpolation between sum (cold rows) and mean (hot rows) by the expected
                # number of updates the row receives per replica and interval
                B = float(merge[3:].replace("bf16", "") or 16)
                T = n_global * 80.0 / syncs
                pv = counts.double() / counts.sum()
                pn = counts.double() ** 0.75
                pn = pn / pn.sum()
                U = 10.5 * T * (pv if ti == 0 else (pv + 5 * pn))
                lam = torch.clamp(B / ((G - 1) * U / G).clamp_min(1e-30), max=1.0).float()
                w = lam + (1 - lam) / G
                if merge.endswith("bf16"):   # deltas travel and are summed as bfloat16 (ring all-reduce in bf16)
                    dl = (stack - bases[ti][None]).bfloat16()
                    acc = dl[0]
                    for r in range(1, G):
                        acc = acc + dl[r]
                    new = bases[ti] + acc.float() * w[:, None]
                else:
                    new = bases[ti] + (stack - bases[ti][None]).sum(0) * w[:, None]
            elif merge == "avg":
                new = stack.mean(0)
            elif merge == "delta":
                new = bases[ti] + (stack - bases[ti][None]).sum(0)
            else:  # sparse_avg, as n2v_hip.sgns.merge_replicas
                delta = stack - bases[ti][None]
                cnt = (delta != 0).any(dim=2).float().sum(0).clamp_min(1.0)
                new = bases[ti] + delta.sum(0) / cnt[:, None]
            for m in models:
                getattr(m, name).copy_(new)
            bases[ti].copy_(new)
    return models[0]


def simulate_tier(G, corpus, n_nodes, rounds, syncs, K, B=256.0, mode="atomic", thr=1.0):
    """Two-tier merge: rows whose expected updates per full interval exceed the budget B (hubs, frequent
    negatives) are merged K times per full interval (small message), all rows once per full interval.
    Weights as merge 'hot', computed for the interval a row actually waited."""
    n = corpus.walks.shape[0] // rounds
    models = [sgns.SgnsModel(n_nodes, dim=128, window=10, negative=5, seed=1, update_mode=mode) for _ in range(G)]
    counts = torch.bincount(corpus.walks.reshape(-1).long(), minlength=n_nodes)
    for m in models:
        m.build_vocab(counts=counts)
    shards = []
    for r in range(G):
        b, e = sgns.shard_bounds(n, G, r)
        idx = (torch.arange(rounds, device=corpus.walks.device)[:, None] * n +
               torch.arange(b, e, device=corpus.walks.device)[None, :]).reshape(-1)
        shards.append((corpus.walks[idx].contiguous(), corpus.lens[idx].contiguous(), b * rounds))
    n_global = corpus.walks.shape[0]
    bases = [models[0].syn0.clone(), models[0].syn1neg.clone()]
    pv = counts.double() / counts.sum()
    pn = counts.double() ** 0.75
    pn = pn / pn.sum()
    T_full = n_global * 80.0 / syncs

    def weight(ti, T):
        U = 10.5 * T * (pv if ti == 0 else (pv + 5 * pn))
        lam = torch.clamp(B / ((G - 1) * U / G).clamp_min(1e-30), max=1.0).float()
        return lam, lam + (1 - lam) / G
    hot = [weight(ti, T_full)[0] < 1.0 / thr for ti in range(2)]   # expected updates above thr x budget
    w_sub = [weight(ti, T_full / K)[1] for ti in range(2)]
    print("   tier: hot rows syn0 %d syn1neg %d of %d" % (int(hot[0].sum()), int(hot[1].sum()), n_nodes), flush=True)
    for c in range(syncs * K):
        for r, m in enumerate(models):
            w, l, off = shards[r]
            b, e = sgns.shard_bounds(w.shape[0], syncs * K, c)
            if e > b:
                m.train_pass(w[b:e], l[b:e], sentences_base=b * G, sentences_step=G, sentences_total=n_global,
                             walk_id_base=off + b)
        full = (c + 1) % K == 0
        for ti, name in enumerate(("syn0", "syn1neg")):
            rows = torch.nonzero(hot[ti] if not full else torch.ones_like(hot[ti])).flatten()
            if rows.numel() == 0:
                continue
            stack = torch.stack([getattr(m, name)[rows] for m in models])
            base = bases[ti][rows]
            wr = torch.where(hot[ti][rows], w_sub[ti][rows], weight(ti, T_full)[1][rows])
            new = base + (stack - base[None]).sum(0) * wr[:, None]
            for m in models:
                getattr(m, name)[rows] = new
            bases[ti][rows] = new
    return models[0]


def _hub_partition(n=20000, k=100, m_in=200000, m_out=40000, seed=0):
    """Degree-corrected planted partition: communities + Pareto node activity (hubs)."""
    rs = np.random.RandomState(seed)
    comm = rs.randint(0, k, n)
    theta = rs.pareto(1.5, n) + 1.0
    order = np.argsort(comm, kind="stable")
    starts = np.searchsorted(comm[order], np.arange(k + 1))
    src, dst = [], []
    per = m_in // k
    for c in range(k):
        members = order[starts[c]:starts[c + 1]]
        if len(members) < 2:
            continue
        pr = theta[members] / theta[members].sum()
        src.append(rs.choice(members, per, p=pr))
        dst.append(rs.choice(members, per, p=pr))
    pr = theta / theta.sum()
    src.append(rs.choice(n, m_out, p=pr))
    dst.append(rs.choice(n, m_out, p=pr))
    src, dst = np.concatenate(src), np.concatenate(dst)
    keep = src != dst
    src, dst = src[keep], dst[keep]
    key = np.minimum(src, dst) * n + np.maximum(src, dst)
    _, first = np.unique(key, return_index=True)
    first.sort()
    return np.stack([src[first], dst[first]], 1)


def setup(kind):
    if kind == "pp":
        return _auc_setup()
    if kind == "hub" or kind.startswith("hub:"):
        from n2v_hip import csr
        from oracle import sgns_oracle
        if kind == "hub":
            edges = _hub_partition()
        else:       # hub:<n>: the same generator scaled up (communities of ~200 nodes, 10 n + 2 n edges)
            nn = int(kind.split(":")[1])
            edges = _hub_partition(n=nn, k=nn // 200, m_in=10 * nn, m_out=2 * nn, seed=2)
        tr, te = sgns_oracle.split_edges(edges)
        g = csr.from_edges(tr[:, 0], tr[:, 1], None, False)
        print("hub graph: nodes %d train edges %d max degree %d" % (g.n_nodes, len(tr), g.degrees.max()), flush=True)
        ing = set(g.labels.tolist())
        te = np.array([e for e in te.tolist() if e[0] in ing and e[1] in ing])
        neg = np.array(sgns_oracle.build_neg_samples(g.labels.tolist(), edges.tolist(), seed=0))
        return g, te, neg
    from n2v_hip import csr, synth
    from oracle import sgns_oracle
    n = int(kind.split(":")[1]) if ":" in kind else 20000
    u, v = synth.barabasi_albert_edges(n, 10, 42)
    edges = np.stack([u, v], 1)
    tr, te = sgns_oracle.split_edges(edges)
    g = csr.from_edges(tr[:, 0], tr[:, 1], None, False)
    ing = set(g.labels.tolist())
    te = np.array([e for e in te.tolist() if e[0] in ing and e[1] in ing])
    neg = np.array(sgns_oracle.build_neg_samples(g.labels.tolist(), edges.tolist(), seed=0))
    return g, te, neg


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "pp"
    g, te, neg = setup(kind)
    Gr = node2vec.Graph.from_csr(g, 1.0, 1.0, rng="philox", seed=1)
    Gr.preprocess_transition_probs()
    rounds = 10
    corpus = Gr.simulate_walks(rounds, 80)
    te_d = np.stack([g.dense_of(te[:, 0]), g.dense_of(te[:, 1])], 1)
    neg_d = np.stack([g.dense_of(neg[:, 0]), g.dense_of(neg[:, 1])], 1)
    from oracle import c_oracle, sgns_oracle
    import time
    counts = np.bincount(corpus.walks.cpu().numpy().reshape(-1), minlength=g.n_nodes)
    si, cum = sgns.vocab_tables(counts, 1e-3)
    for thr in ((1, 64) if kind == "pp" else (64,)):
        syn0, syn1 = c_oracle.sgns_init(g.n_nodes, 128, 128, 1)
        t = time.time()
        c_oracle.sgns_train(corpus.walks.cpu().numpy(), corpus.lens.cpu().numpy(), syn0, syn1, 128, 10, 5, si, cum,
                            n_threads=thr)
        auc, ap = linkpred.get_roc_score(torch.from_numpy(syn0).cuda(), te_d, neg_d)
        print("CPU comparator threads=%d: AUC %.5f AP %.5f (%.0fs)" % (thr, auc, ap, time.time() - t), flush=True)
    m = simulate(1, corpus, g.n_nodes, rounds, "avg", 1)
    print("G=1: AUC %.5f" % linkpred.get_roc_score(m.vectors(), te_d, neg_d)[0], flush=True)
    auto = {G: sgns.auto_syncs(corpus.walks.shape[0] * 80, g.n_nodes, G) for G in (2, 8)}
    print("auto syncs:", auto, flush=True)
    sync_list = [int(x) for x in os.environ.get("SYNCS", "4,16,64").split(",")]
    for G in [int(x) for x in os.environ.get("GS", "2,8").split(",")]:
        for merge in os.environ.get("MERGES", "sparse_avg,avg,delta").split(","):
            for syncs in sync_list:
                if syncs <= 0:
                    syncs = max(1, auto[G] // (-syncs if syncs < 0 else 1))   # 0: auto, -k: auto/k
                if merge.startswith("tier"):
                    k, _, thr = merge[4:].partition("x")     # tier8x2: 8 sub-merges, rows above 2x the budget
                    m = simulate_tier(G, corpus, g.n_nodes, rounds, syncs, int(k or 8), thr=float(thr or 1))
                else:
                    m = simulate(G, corpus, g.n_nodes, rounds, merge, syncs)
                auc, ap = linkpred.get_roc_score(m.vectors(), te_d, neg_d)
                print("G=%d merge=%-10s syncs=%3d: AUC %.5f AP %.5f" % (G, merge, syncs, auc, ap), flush=True)


if __name__ == "__main__":
    main()
