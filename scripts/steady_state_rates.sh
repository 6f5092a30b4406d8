#!/bin/bash
# Dev helper (GPU box): C5 and its 1/8 shares WITHOUT a run's fill and drain (scripts/wall_rate.py, MRT_STEADY: frames ending between
# two calls that both found the pipeline full) -> the strong-scaling projection of DESIGN.md 7.  -> gpurun_out/r05o/steady.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05o; mkdir -p $O
( MRT_STEADY=1 MRT_HINT=2,2 MRT_WARMUP=8 python scripts/wall_rate.py stress 1920 1080 4096 36
  MRT_STEADY=1 MRT_HINT=4,2 MRT_WARMUP=8 python scripts/wall_rate.py stress 1920 1080 4096 36
  MRT_STEADY=1 MRT_SHARD=0,8 MRT_WARMUP=32 python scripts/wall_rate.py stress 1920 1080 4096 160
  MRT_STEADY=1 MRT_SHARD=5,8 MRT_WARMUP=32 python scripts/wall_rate.py stress 1920 1080 4096 160
  MRT_STEADY=1 MRT_SHARD=3,8 MRT_WARMUP=32 python scripts/wall_rate.py stress 1920 1080 4096 160 ) 2>&1 | grep -v amdgpu | sed -e 's/HIER=None BOXES=None RNG=None//' -e 's/, schedule {.*//' | tee $O/steady.txt
