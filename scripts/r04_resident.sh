#!/bin/bash
# Dev helper (GPU box): launch widths as shares of the waves the chip HOLDS for the scene (MRT_EXP_RESIDENT) vs of n_waves
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$1; mkdir -p $O
for e in "" 1; do
  ( [ -n "$e" ] && export MRT_EXP_RESIDENT=1
    MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 24 0
    MRT_WARMUP=32 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 4 16 0
    MRT_WARMUP=28 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 12 0
    MRT_WARMUP=6 python scripts/wall_rate.py stress 1920 1080 512 12
    MRT_WARMUP=4 python scripts/wall_rate.py stress 1920 1080 4096 3
    for n in 36 70; do MRT_WARMUP=8 python scripts/wall_rate.py stress$n 1920 1080 64 24; done ) 2>&1 | grep -v amdgpu.ids | cut -c1-130 | sed "s/^/resident=$e /"
done | tee $O/rates.txt
