# A/B of the on-the-fly walk's draw-before-build paths on C3 (tools/hybrid_probe.py); N2V_OTF_DRAW_FIRST caps the mode.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/otf_ab
: > gpurun_out/otf_ab/otf_ab.log
for m in ${MODES:-0 1 2}; do
  echo "== N2V_OTF_DRAW_FIRST=$m" >> gpurun_out/otf_ab/otf_ab.log
  N2V_OTF_DRAW_FIRST=$m timeout -k 10 280 python3 tools/hybrid_probe.py C3 >> gpurun_out/otf_ab/otf_ab.log 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/otf_ab/otf_ab.log | cut -c1-220
