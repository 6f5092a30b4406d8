#!/bin/bash
# Dev helper (GPU box): large scenes of several sizes and C5's 1/2 share under a half / a quarter / an eighth x 2 -> gpurun_out/r05n/sweep3.txt
# (appended to profiles/r05_schedule_sweep.txt: what the starting rule for large scenes rests on)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05n; mkdir -p $O
run() { MRT_SHARD=$1 MRT_HINT=$2 MRT_WARMUP=$3 timeout -k 10 300 python scripts/wall_rate.py $5 $6 $7 $8 $4 2>&1 | grep -v amdgpu.ids | sed -e 's/HIER=None BOXES=None RNG=None//' -e "s/^/shard $1 hint $2: /" | cut -c1-112; }
( for H in 2,2 4,2 8,2; do MRT_SHARD= run "" $H 24 60 stress 1920 1080 512; done
  for S in stress70 stress50 stress36; do for H in 2,2 4,2 8,2; do MRT_SHARD= run "" $H 24 120 $S 1920 1080 64; done; done
  for H in 2,2 4,2 8,2; do run 0,2 $H 24 32 stress 1920 1080 4096; done ) > $O/sweep3.txt 2>&1
cat $O/sweep3.txt
