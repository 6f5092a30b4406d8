#!/bin/bash
# Dev helper (GPU box): C5's shares under the launch-width controller, decisions traced
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$1; mkdir -p $O
( MRT_TRACE_WIDTH=1 MRT_WARMUP=24 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 10 0
  MRT_TRACE_WIDTH=1 MRT_WARMUP=24 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 4 12 0
  MRT_TRACE_WIDTH=1 MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 32 0
  MRT_TRACE_WIDTH=1 MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py cover-glass 1920 1080 512 0 8 32 0 ) 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee $O/shards.txt
