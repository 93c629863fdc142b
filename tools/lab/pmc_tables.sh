#!/bin/bash
# PMC traffic of the table builder alone (C3): WRITE_SIZE and FETCH_SIZE in separate passes (never with other trace
# domains).  Usage on the GPU box: bash tools/lab/pmc_tables.sh <outdir>
set -e -o pipefail
O=${1:-gpurun_out/pmc_tables}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 tools/lab/preprocess_ab.py > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 tools/lab/preprocess_ab.py > $O/fetch.log 2>&1
python3 - "$O" <<'P'
import csv, glob, sys, collections
O = sys.argv[1]
for sub, c in (("write", "WRITE_SIZE"), ("fetch", "FETCH_SIZE")):
    f = max(glob.glob(O + "/" + sub + "/**/*counter_collection.csv", recursive=True), key=lambda p: __import__("os").path.getsize(p))
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0][-60:]
        acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in sorted(acc, key=acc.get, reverse=True)[:6]:
        print("%s %-62s %8.2f GB per launch (%d launches)" % (c, k, acc[k] * 1024 / len(n[k]) / 1e9, len(n[k])))
P
