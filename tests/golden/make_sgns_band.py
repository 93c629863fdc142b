"""Comparator fixtures of the SGNS acceptance band (BASELINE.json: link-prediction AUC within +-0.002).

CPU only, run in the build container (`python tests/golden/make_sgns_band.py [case ...]`; the 131 072-node cases take
17 and 63 minutes on one core each, the 400k-node case 3.8 h; cases run in parallel processes).  For every case of tests/band_cases.py:

  1. the C oracle (oracle/n2v_oracle.c — pinned bit for bit to the reference's src/node2vec.py by tests/golden/*.npz)
     walks the training graph with Philox uniforms, seed 1: the same walks the HIP kernel produces (the -m gpu tests
     check the hash stored here before they train);
  2. the SEQUENTIAL comparator oracle/sgns_oracle.c (one thread; gensim 3.2.0's published algorithm with the
     arguments of src/main.py:82-90 — parity unpinned, see its header) trains d = 128, window 10, 5 negatives, seed 1;
  3. cosine link-prediction AUC / AP on the held-out half of the edges against sampled non-edges
     (src/main_link.py:173-204,525-563), fp64 numpy + sklearn.

Stored per case (small JSON under tests/golden/sgns_band/): graph / walk / count hashes, pair count, AUC, AP.
Nothing of the reference is read at run time; no GPU is used."""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def cpu_auc(syn0, pos, neg):
    from sklearn.metrics import average_precision_score, roc_auc_score
    v = syn0.astype(np.float64)
    v = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-300)
    sp = np.einsum("ij,ij->i", v[pos[:, 0]], v[pos[:, 1]])
    sn = np.einsum("ij,ij->i", v[neg[:, 0]], v[neg[:, 1]])
    y = np.concatenate([np.ones(len(sp)), np.zeros(len(sn))])
    s = np.concatenate([sp, sn])
    return float(roc_auc_score(y, s)), float(average_precision_score(y, s))


def make(name):
    import band_cases
    from oracle import c_oracle, sgns_oracle
    t0 = time.time()
    case = band_cases.build(name)
    g, rounds, L = case["graph"], case["rounds"], case["L"]
    co = c_oracle.CsrOracle(g.row_ptr, g.col, g.w, 1.0, 1.0)
    co.preprocess(first_order_shortcut=True)          # p = q = 1: every (src, dst) table is dst's node table
    walks, lens, _ = co.walk(g.start_order, rounds, L, mode="philox", seed=1)
    counts = np.bincount(walks[walks >= 0], minlength=g.n_nodes)
    si, cum = sgns_oracle.vocab_tables(counts, 1e-3)
    syn0, syn1 = c_oracle.sgns_init(g.n_nodes, 128, 128, 1)
    t1 = time.time()
    pairs = c_oracle.sgns_train(walks, lens, syn0, syn1, 128, 10, 5, si, cum, n_threads=1)
    t2 = time.time()
    auc, ap = cpu_auc(syn0, case["te_d"], case["neg_d"])
    out = {
        "case": name, "n_nodes": int(g.n_nodes), "nnz": int(g.nnz), "max_degree": int(g.degrees.max()),
        "rounds": rounds, "walk_length": L, "walk_seed": 1, "sgns_seed": 1, "dim": 128, "window": 10, "negative": 5,
        "edges_sha16": case["edges_sha"], "walks_sha16": band_cases.sha16(walks), "lens_sha16": band_cases.sha16(lens),
        "counts_sha16": band_cases.sha16(counts.astype(np.int64)), "n_tokens": int(lens.sum()),
        "n_test_pairs": int(len(case["te_d"])), "n_neg_pairs": int(len(case["neg_d"])),
        "comparator": "oracle/sgns_oracle.c, 1 thread (sequential)", "pairs_cpu": int(pairs),
        "auc_cpu": auc, "ap_cpu": ap, "comparator_seconds": round(t2 - t1, 1),
    }
    os.makedirs(band_cases.BAND_DIR, exist_ok=True)
    with open(band_cases.fixture_path(name), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("%s: AUC %.5f AP %.5f pairs %d (%.0f s total, comparator %.0f s)" % (name, auc, ap, pairs, time.time() - t0, t2 - t1),
          flush=True)
    return out


if __name__ == "__main__":
    import band_cases
    names = sys.argv[1:] or list(band_cases.CASES)
    from oracle import c_oracle
    c_oracle.build()
    with mp.get_context("spawn").Pool(min(len(names), 4)) as pool:
        pool.map(make, names)
