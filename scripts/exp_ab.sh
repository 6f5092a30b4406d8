#!/bin/bash
# Dev helper (GPU box): wall-clock rates of the regular library and of every myraytracer_amd/lib/alt_*.so (scripts/exp_build.sh)
#   usage: scripts/exp_ab.sh <out tag> [c3|c5|both]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/ab_$1; mkdir -p $O
for rep in 1 2; do
for f in myraytracer_amd/lib/libmyraytracer_amd.so myraytracer_amd/lib/alt_*.so; do
  t=$(basename $f .so)
  if [ "$2" != "c5" ]; then echo -n "$t: "; MRT_WARMUP=4 MRT_LIB_OVERRIDE=$PWD/$f python scripts/wall_rate.py cover-glass 1920 1080 512 12 2>/dev/null | tail -1; fi
  if [ "$2" != "c3" ]; then echo -n "$t: "; MRT_WARMUP=6 MRT_LIB_OVERRIDE=$PWD/$f python scripts/wall_rate.py stress 1920 1080 512 10 2>/dev/null | tail -1; fi
done
done | tee $O/rates.txt
