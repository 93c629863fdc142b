#!/bin/bash
# Diagnostic build of the library with per-phase cycle stamps in the edge-table builder (N2V_TAB_STAMPS), into
# tools/lab/libn2v_hip_stamps.so; run tools/lab/tab_stamps_probe.py with N2V_HIP_LIB pointing at it.
set -e
cd "$(dirname "$0")/../../node2vec-by-ecc_amd/csrc"
OUT=../../tools/lab/libn2v_hip_stamps.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include \
  -DN2V_TAB_STAMPS -shared -o $OUT n2v_walk.hip n2v_walk_fat.hip n2v_walk_otf.hip n2v_mt19937.hip n2v_alias.hip \
  n2v_tables.hip n2v_sgns.hip n2v_merge.hip n2v_bine.hip n2v_sim.hip
echo built $OUT
