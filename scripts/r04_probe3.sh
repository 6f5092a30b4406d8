#!/bin/bash
# Dev helper (GPU box), round 4: does the number of HIP hardware queues bound the frames in flight of a pixel-starved shard?
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe3; mkdir -p $O
for q in 4 8 16; do
for cfg in "8 2" "8 4" "6 4"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$q MRT_NOBATCH=1 MRT_SLOTS=$1 MRT_WAVES_PER_CU=$2 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 $((2*$1)) 0 2>/dev/null | sed "s/^/hwq=$q slots=$1 wpc=$2 /" >> $O/shard_sweep.txt
done
done
cat $O/shard_sweep.txt
