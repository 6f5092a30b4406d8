#!/bin/bash
# Dev helper (HERE, then the GPU box): build kernels.hip under alternative scheduler flags into myraytracer_amd/lib/alt_<tag>.so
# (gitignored), then `gpurun -- bash scripts/flag_ab.sh run` measures C3 wall rate with each.
set -e
BASE="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-vectorize -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form"
SRCS="myraytracer_amd/csrc/kernels.hip myraytracer_amd/csrc/tile_order.hip myraytracer_amd/csrc/api.cpp myraytracer_amd/csrc/multi_gpu.cpp myraytracer_amd/csrc/scenes.cpp myraytracer_amd/csrc/image_io.cpp"
if [ "$1" = "run" ]; then
  for f in myraytracer_amd/lib/libmyraytracer_amd.so myraytracer_amd/lib/alt_*.so; do
    echo -n "$(basename $f): "; MRT_LIB_OVERRIDE=$PWD/$f python scripts/wall_rate.py cover-glass 1920 1080 512 8 2>/dev/null | tail -1
  done
  exit 0
fi
i=0
while read -r tag flags; do
  [ -z "$tag" ] && continue
  ( /opt/rocm/bin/hipcc $BASE $flags -shared -o myraytracer_amd/lib/alt_$tag.so $SRCS -ldl 2>&1 | grep -E "error" || true
    python3 scripts/check_isa.py --flags "${BASE/ -fPIC/} $flags" 2>&1 | tail -1 | sed "s/^/$tag: /" ) &
  i=$((i+1)); [ $((i % 4)) -eq 0 ] && wait
done <<LIST
maxilp -mllvm -amdgpu-sched-strategy=max-ilp
maxmem -mllvm -amdgpu-sched-strategy=max-memory-clause
iterilp -mllvm -amdgpu-sched-strategy=iterative-ilp
bias10 -mllvm -amdgpu-schedule-metric-bias=10
bias90 -mllvm -amdgpu-schedule-metric-bias=90
relaxed -mllvm -amdgpu-schedule-relaxed-occupancy
O2 -O2
LIST
wait
ls -la myraytracer_amd/lib/
