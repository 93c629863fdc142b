"""Two ranks of the drop-in main.py under torch.distributed.run (run with -m gpu).  The box has one
GPU, so both ranks share it and the merges go over gloo through host memory — the rehearsal path
of n2v_hip.dist; on a multi-GPU node the same code runs one rank per GPU over RCCL."""
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import ROOT, load_case

pytestmark = pytest.mark.gpu


def test_two_rank_main_writes_one_merged_embedding(tmp_path):
    z = load_case("karate_p1_q1")
    edgelist = tmp_path / "karate.edgelist"
    edgelist.write_text("".join("%d %d\n" % (u, v) for u, v in z["edges"].tolist()))
    (tmp_path / "emb").mkdir()
    out = tmp_path / "emb" / "karate.emb"
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "node2vec-by-ecc_amd", "main.py"),
           "--input", str(edgelist), "--output", str(out), "--dimensions", "64", "--walk-length", "20",
           "--num-walks", "6", "--rng", "philox", "--seed", "3"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = out.read_text().splitlines()
    assert lines[0] == "34 64" and len(lines) == 35
    vec = np.array([[float(x) for x in l.split()[1:]] for l in lines[1:]])
    assert np.isfinite(vec).all() and np.abs(vec).max() > 1e-3
    counts_total = 34 * 6 * 20   # every token of every rank's walks was counted once
    assert {l.split()[0] for l in lines[1:]} == {str(i) for i in range(1, 35)} and counts_total > 0
