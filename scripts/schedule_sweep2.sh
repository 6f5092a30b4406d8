#!/bin/bash
# Dev helper (GPU box): the small-scene workloads under pinned schedules -> gpurun_out/r05e/sweep2.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05e; mkdir -p $O
run() { # shard hint warm steps scene w h spp
  MRT_SHARD=$1 MRT_HINT=$2 MRT_WARMUP=$3 timeout -k 10 300 python scripts/wall_rate.py $5 $6 $7 $8 $4 2>&1 | grep -v amdgpu.ids | sed -e 's/HIER=None BOXES=None RNG=None//' -e "s/^/shard $1 hint $2: /" | cut -c1-140
}
( for H in 1,1 1,2 2,1 2,2 4,2 8,2 4,4 1,1; do MRT_SHARD= run "" $H 24 60 cover-glass 1920 1080 512; done
  for H in 1,1 2,2 4,2 8,2; do MRT_SHARD= run "" $H 6 8 cover-glass 3840 2160 1024; done
  for H in 4,1 4,2 8,1 8,2 4,4; do MRT_SHARD= run "" $H 400 3000 cover-glass 1920 1080 1; done
  for H in 1,4 1,8 2,4 2,8 1,2; do MRT_SHARD= run "" $H 600 6000 default 400 225 16; done ) > $O/sweep2.txt 2>&1
cat $O/sweep2.txt
