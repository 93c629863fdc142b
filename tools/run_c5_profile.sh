#!/bin/bash
# GPU-box recipe behind profiles/r01/bine_c5_*: the C5 bench line, its rocprofv3 kernel stats and the two PMC
# passes (FETCH_SIZE / WRITE_SIZE need separate runs).  Usage: gpurun -- 'bash tools/run_c5_profile.sh'
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
timeout -k 10 300 python bench.py --config C5 --steps 5 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5_kt -- python3 bench.py --config C5 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_c5_prof.json 2> $O/bench_c5_prof.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/prof_c5_fetch -- python3 bench.py --config C5 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5_fetch.json 2> $O/bench_c5_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/prof_c5_write -- python3 bench.py --config C5 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_c5_write.json 2> $O/bench_c5_write.err
tail -3 $O/bench_c5.err
cut -c1-400 $O/bench_c5.json
