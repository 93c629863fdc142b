"""Turns the raw rocprofv3 outputs of tools/run_c3_profile.sh into the summaries kept under profiles/rNN/:
  kernel_stats.csv  the top rows of the --kernel-trace --stats summary,
  pmc_raw.json      FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports them) summed per kernel, with dispatch counts,
  traffic.json      HBM bytes PER LAUNCH of the walk / SGNS / table kernels after the corrections calibrated on this
                    chip (profiles/r01/fetch_write_calibration.json, MI355X_MICROARCH.md): FETCH_SIZE x2 for
                    streaming row reads of >= 16 B per lane (sgns_kernel, merge kernels), x1 for the walk's gathers
                    (one 64-B request per 32-B slot), WRITE_SIZE x1.
Usage: python tools/collect_profile.py <raw dir with kt/ fetch/ write/> <out dir>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

raw, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)


def find(sub, suffix):
    hits = glob.glob(os.path.join(raw, sub, "**", "*" + suffix), recursive=True)
    return max(hits, key=os.path.getsize) if hits else None


stats = find("kt", "kernel_stats.csv")
if stats:
    rows = list(csv.reader(open(stats)))
    with open(os.path.join(out, "kernel_stats.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows[:26])


def short(name):
    for key in ("sgns_kernel", "sgns_shared_kernel", "walk_fat2_kernel", "walk_fat_kernel", "walk_kernel",
                "edge_tables_wave_kernel", "edge_tables_kernel", "fat_expand_kernel", "node_tables_kernel",
                "edge_recs_kernel", "merge_snapshot_kernel", "merge_hot_apply_kernel", "merge_flush_kernel"):
        if key in name:
            return key
    return None


pmc = {}
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = find(sub, "counter_collection.csv")
    if not f:
        continue
    acc, n = defaultdict(float), defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        if k is None or row["Counter_Name"] != counter:
            continue
        acc[k] += float(row["Counter_Value"])
        n[k].add(row["Dispatch_Id"])
    pmc[counter] = {k: {"kb": acc[k], "dispatches": len(n[k])} for k in acc}
json.dump(pmc, open(os.path.join(out, "pmc_raw.json"), "w"), indent=1)

FETCH_X = {"sgns_kernel": 2.0, "merge_snapshot_kernel": 2.0, "merge_hot_apply_kernel": 2.0, "merge_flush_kernel": 2.0}
traffic = {}
for k in set(pmc.get("FETCH_SIZE", {})) | set(pmc.get("WRITE_SIZE", {})):
    fe, wr = pmc.get("FETCH_SIZE", {}).get(k), pmc.get("WRITE_SIZE", {}).get(k)
    if not fe or not wr:
        continue
    fetch = fe["kb"] * 1024 * FETCH_X.get(k, 1.0) / fe["dispatches"]
    write = wr["kb"] * 1024 / wr["dispatches"]
    traffic[k] = {"fetch_per_launch": fetch, "write_per_launch": write, "bytes_per_launch": fetch + write,
                  "fetch_correction": FETCH_X.get(k, 1.0)}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["bytes_per_launch"] / 1e9, 3) for k, v in traffic.items()}))
