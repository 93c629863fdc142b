# Builds A/B variants of the on-the-fly walk kernel into tools/lab/_ab/ (run in the build container; the .so files travel).
set -e
cd "$(dirname "$0")/../../node2vec-by-ecc_amd/csrc"
make -s
mkdir -p ../../tools/lab/_ab
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include"
OTHERS=$(ls _obj/*.o | grep -v n2v_walk_otf.o)
build() { name=$1; shift; /opt/rocm/bin/hipcc $FLAGS "$@" -c n2v_walk_otf.hip -o /tmp/otf_$name.o && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/lab/_ab/lib_$name.so /tmp/otf_$name.o $OTHERS; }
build accept -DN2V_OTF_LAB_ALWAYS_ACCEPT
build halfgrid -DN2V_OTF_LAB_GRID_DIV=2
build build_always -DN2V_OTF_LAB_DRAW_FIRST=0
build sum_always -DN2V_OTF_LAB_DRAW_FIRST=1
ls -la ../../tools/lab/_ab
