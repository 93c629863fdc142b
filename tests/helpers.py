"""Shared helpers for the parity tests: load golden fixtures, rebuild oracle graphs."""
import glob
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

GRAPH_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz"))
                     if os.path.basename(p) != "alias_setup.npz")


def load_case(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def case_weights(z):
    w = z["weights"]
    return [int(x) for x in w] if bool(z["int_weights"]) else [float(x) for x in w]


def oracle_graph(z):
    from oracle.n2v_oracle import OracleGraph
    return OracleGraph([tuple(e) for e in z["edges"].tolist()], case_weights(z), bool(z["directed"]))


def golden_walks(z, i):
    flat, ptr = z["walks_%d_flat" % i], z["walks_%d_ptr" % i]
    return [flat[ptr[k]:ptr[k + 1]].tolist() for k in range(len(ptr) - 1)]


def walks_from_padded(walks, lens, labels=None):
    out = []
    for row, n in zip(np.asarray(walks), np.asarray(lens)):
        r = row[:n]
        out.append((labels[r] if labels is not None else r).tolist())
    return out
