"""Scratch probe (not a test): SGNS throughput on walks of a BASELINE config, per sharing mode.

    MODES=plain,agent,atomic python tools/sgns_probe.py C3 [rounds] [dim]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "node2vec-by-ecc_amd"))
import torch

import node2vec
from n2v_hip import sgns, synth


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C2"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    dim = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    cg, info = synth.make_config_graph(name)
    p, q = (1.0, 1.0) if name == "C2" else (0.25, 4.0)
    g = node2vec.Graph.from_csr(cg, p, q, rng="philox", seed=1)
    g.preprocess_transition_probs()
    corpus = g.simulate_walks(rounds, 80)
    torch.cuda.synchronize()
    print(name, "walks", tuple(corpus.walks.shape), flush=True)
    for mode in os.environ.get("MODES", "plain,agent,atomic").split(","):
        m = sgns.SgnsModel(cg.n_nodes, dim=dim, window=10, negative=5, seed=1, update_mode=mode.split("+")[0], share_negatives=mode.endswith("+share"))
        m.build_vocab(corpus.walks)
        for rep in range(2):
            m.pair_count.zero_()
            torch.cuda.synchronize()
            t = time.perf_counter()
            sgns.train(m, corpus.walks, corpus.lens, epochs=1)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            n = m.pairs_trained()
            print("%-7s pass %d: %.3fs  %d pairs  %.3e pairs/s  algorithmic %.2f TB/s" % (
                mode, rep, dt, n, n / dt, n / dt * 7168 * (m.stride / 128) / 1e12), flush=True)
        print(mode, "finite:", bool(torch.isfinite(m.syn0).all()), "syn0 absmax %.3f" % float(m.syn0.abs().max()),
              flush=True)
        del m


if __name__ == "__main__":
    main()
