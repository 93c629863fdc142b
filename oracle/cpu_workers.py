"""Worker of bench.py's multi-process CPU baseline — TEST/BENCH INFRASTRUCTURE ONLY.

The reference fans its walks out over processes by contiguous blocks of start nodes
(src/main_link.py:259-292); this is the same split for the pure-Python restatement.  Arrays come from
.npy files (memory-mapped) so that spawning does not pickle the graph per worker."""
import time

import numpy as np


def walk_block(args):
    path, directed, p, q, lo, hi, budget_s, seed = args
    from oracle import n2v_oracle as orc
    z = {k: np.load("%s_%s.npy" % (path, k), mmap_mode="r") for k in ("labels", "row_ptr", "col", "start_order")}
    G = orc.CsrBackedGraph(z["labels"], z["row_ptr"], z["col"], None, z["start_order"], directed)
    o = orc.Node2VecOracle(G, directed, p, q)
    rs = np.random.RandomState(seed)
    t0 = time.perf_counter()
    steps = done = 0
    for node in G.nodes[lo:hi]:
        steps += len(o.node2vec_walk(80, node, rs.random_sample, on_the_fly=True)) - 1
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return steps, done, time.perf_counter() - t0


if __name__ == "__main__":   # python -m oracle.cpu_workers <path> <directed> <p> <q> <lo> <hi> <budget_s> <seed>
    import json
    import sys
    a = sys.argv[1:]
    print(json.dumps(walk_block((a[0], a[1] == "1", float(a[2]), float(a[3]), int(a[4]), int(a[5]), float(a[6]), int(a[7])))))
