#!/bin/bash
# Dev helper (GPU box): C5 under other hierarchy depths / top sizes (MRT_HIER=max_levels,top_target), boxes in LDS as they fit
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$1; mkdir -p $O
for h in 4,0 4,64 3,256 3,64 4,16; do
  MRT_HIER=$h MRT_WARMUP=6 python scripts/wall_rate.py stress 1920 1080 512 10 2>/dev/null | tee -a $O/rates.txt
done
for h in 4,0 3,64 2,256; do
  MRT_HIER=$h python scripts/wall_rate.py stress70 1920 1080 64 8 2>/dev/null | tee -a $O/rates.txt
  MRT_HIER=$h python scripts/wall_rate.py stress36 1920 1080 64 8 2>/dev/null | tee -a $O/rates.txt
done
