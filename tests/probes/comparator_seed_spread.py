"""How much does the sequential comparator's AUC move with its seed alone?  The +-0.002 band compares ONE realisation
of the reference algorithm's randomness (sub-sampling, window shrink, negatives: gensim's LCG flow in the oracle) with
ONE realisation of the kernel's (hash-keyed streams); both are draws from the same distribution of outcomes.  CPU only:
python tests/probes/comparator_seed_spread.py [case] [seeds...]"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def run(args):
    name, seed, init_seed = args
    import band_cases
    from make_sgns_band import cpu_auc
    from oracle import c_oracle, sgns_oracle
    case = band_cases.build(name)
    g = case["graph"]
    co = c_oracle.CsrOracle(g.row_ptr, g.col, g.w, 1.0, 1.0)
    co.preprocess(first_order_shortcut=True)
    walks, lens, _ = co.walk(g.start_order, case["rounds"], case["L"], mode="philox", seed=1)
    counts = np.bincount(walks[walks >= 0], minlength=g.n_nodes)
    si, cum = sgns_oracle.vocab_tables(counts, 1e-3)
    syn0, syn1 = c_oracle.sgns_init(g.n_nodes, 128, 128, init_seed)
    t = time.time()
    c_oracle.sgns_train(walks, lens, syn0, syn1, 128, 10, 5, si, cum, n_threads=1, seed=seed)
    auc, ap = cpu_auc(syn0, case["te_d"], case["neg_d"])
    print("%s training seed %d init seed %d: AUC %.5f AP %.5f (%.0f s)" % (name, seed, init_seed, auc, ap, time.time() - t), flush=True)
    return auc


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "hub20k_10x80"
    seeds = [int(x) for x in sys.argv[2:]] or [1, 2, 3, 4, 5, 6]
    from oracle import c_oracle
    c_oracle.build()
    jobs = [(name, s, 1) for s in seeds] + [(name, 1, 2), (name, 1, 3)]
    with mp.get_context("spawn").Pool(min(len(jobs), 7)) as pool:
        aucs = pool.map(run, jobs)
    a = np.array(aucs[:len(seeds)])
    print("%s: training-seed spread over %d seeds (same walks, same initial tables): mean %.5f std %.5f min %.5f max %.5f" % (
        name, len(seeds), a.mean(), a.std(ddof=1), a.min(), a.max()))
    print("%s: initial-table seeds 1,2,3 (training seed 1): %s" % (name, ["%.5f" % x for x in [aucs[0]] + aucs[len(seeds):]]))
