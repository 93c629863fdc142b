"""The bench.py contract on a small config (run with -m gpu): one JSON line with the fields
the driver reads, the roofline and cpu_baseline objects, sane magnitudes."""
import json
import os
import subprocess
import sys

import pytest

from helpers import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_contract_c2():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "C2", "--rounds", "2",
                          "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 1 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["vs_baseline"] is None and j["data"] == "synthetic" and "workload" in j["config"]
    assert j["value"] > 1e9 and j["sgns"]["value"] > 1e7
    r = j["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 2
    c = j["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "walk-steps/s"
    assert j["walk"]["steps_per_step_global"] == 100000 * 2 * 79
