#!/bin/bash
# Dev helper (HERE): build kernels.hip with extra -D flags into myraytracer_amd/lib/alt_<tag>.so (gitignored), linking the other
# objects of the regular build.   usage: scripts/exp_build.sh <tag> <flags...>      (then: MRT_LIB_OVERRIDE=.../alt_<tag>.so)
set -e
tag=$1; shift
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-vectorize -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form"
mkdir -p build/exp
/opt/rocm/bin/hipcc $FLAGS "$@" -x hip -c -o build/exp/kernels_$tag.o myraytracer_amd/csrc/kernels.hip
OTHERS=$(ls build/obj/*.o | grep -v "^build/obj/kernels.hip.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o myraytracer_amd/lib/alt_$tag.so build/exp/kernels_$tag.o $OTHERS build/obj/build_id.cpp -ldl
python3 scripts/check_isa.py --flags "${FLAGS/ -fPIC/} $*" 2>&1 | tail -1 | sed "s/^/$tag: /"
