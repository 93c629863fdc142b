# GPU box: product library, then the A/B builds named in VARIANTS (tools/lab/_ab/lib_<name>.so), on C3 via otf_quick.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && mkdir -p gpurun_out/otf_ab
: > gpurun_out/otf_ab/variants.log
timeout -k 10 200 python3 tools/lab/otf_quick.py >> gpurun_out/otf_ab/variants.log 2>&1 || exit 1
for v in ${VARIANTS:-lds128 accept}; do
  N2V_HIP_LIB=$PWD/tools/lab/_ab/lib_$v.so timeout -k 10 200 python3 tools/lab/otf_quick.py >> gpurun_out/otf_ab/variants.log 2>&1 || exit 1
done
grep -v amdgpu.ids gpurun_out/otf_ab/variants.log
