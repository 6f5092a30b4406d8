#!/bin/bash
# Dev helper (GPU box): pipelined rate of the large-scene workloads under pinned schedules (div,mult): which setting should the
# controller arrive at?  -> gpurun_out/r05e/sweep.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05e; mkdir -p $O
run() { # shard hint warm steps scene w h spp
  MRT_SHARD=$1 MRT_HINT=$2 MRT_WARMUP=$3 timeout -k 10 300 python scripts/wall_rate.py $5 $6 $7 $8 $4 2>&1 | grep -v amdgpu.ids | sed -e 's/HIER=None BOXES=None RNG=None//' -e "s/^/shard $1 hint $2: /" | cut -c1-200
}
( for H in 2,1 2,2 4,1 4,2 1,2; do MRT_SHARD= run "" $H 8 6 stress 1920 1080 4096; done
  for H in 4,1 4,2 8,1 8,2 2,2; do run 0,2 $H 16 16 stress 1920 1080 4096; done
  for H in 8,2 8,1 4,2 4,4 4,1; do run 0,4 $H 32 32 stress 1920 1080 4096; done
  for H in 8,2 8,1 4,2; do run 0,8 $H 32 48 stress 1920 1080 4096; done
  for H in 1,1 2,1 4,1 4,2; do run 0,8 $H 8 16 cover-glass 3840 2160 1024; done
  for H in 4,1 4,2 8,1 8,2; do MRT_SHARD= run "" $H 60 300 cover 1200 675 64; done ) > $O/sweep.txt 2>&1
cat $O/sweep.txt
