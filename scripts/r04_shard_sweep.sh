#!/bin/bash
# Dev helper (GPU box): C5's 1/8 share (stream RNG, one mrt_redraw per frame) at fixed frames in flight x waves per CU and launch
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$1; mkdir -p $O
( MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 24 0
  for cfg in "8 2" "8 3" "8 4" "6 3" "4 4" "8 1"; do set -- $cfg
    echo -n "slots $1 waves/CU $2: "; MRT_SLOTS=$1 MRT_WAVES_PER_CU=$2 MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 24 0
  done ) 2>&1 | grep -v amdgpu.ids | cut -c1-150 | tee $O/sweep.txt
