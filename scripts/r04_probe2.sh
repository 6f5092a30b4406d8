#!/bin/bash
# Dev helper (GPU box), round 4: pixel-starved C5 shard (1/8, stream mode, one mrt_redraw per frame): frames in flight x waves per CU and launch
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe2; mkdir -p $O
for cfg in "2 0" "4 8" "4 4" "8 4" "8 2" "6 4" "8 6"; do
  set -- $cfg
  MRT_NOBATCH=1 MRT_SLOTS=$1 MRT_WAVES_PER_CU=$2 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 $((2*$1)) 0 2>/dev/null | sed "s/^/slots=$1 wpc=$2 /" >> $O/shard_sweep.txt
done
cat $O/shard_sweep.txt
