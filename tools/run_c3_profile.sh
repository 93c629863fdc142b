#!/bin/bash
# GPU-box recipe behind profiles/rNN/bench_c3_n1_* and the C3 entries of profiles/traffic.json: the default bench
# line, the same command under rocprofv3 --kernel-trace --stats, and the two PMC passes (FETCH_SIZE / WRITE_SIZE need
# separate runs; never combined with other trace domains).  tools/collect_profile.py turns the raw outputs into
# the committed summaries.  Usage: gpurun --timeout 1100 -- 'bash tools/run_c3_profile.sh r02'
set -e -o pipefail
R=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_c3_$R
mkdir -p $O
FAST="--no-cpu-baseline --no-shared-negatives --no-reference-exact"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 2 --warmup 1 $FAST > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
echo "[profile] kernel trace done" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 bench.py --steps 1 --warmup 0 $FAST > $O/bench_fetch.json 2> $O/bench_fetch.err
echo "[profile] FETCH_SIZE pass done" >&2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 bench.py --steps 1 --warmup 0 $FAST > $O/bench_write.json 2> $O/bench_write.err
echo "[profile] WRITE_SIZE pass done" >&2
python3 tools/collect_profile.py $O $O/summary
tail -2 $O/bench_under_rocprof.err
