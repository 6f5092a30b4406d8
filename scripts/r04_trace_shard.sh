#!/bin/bash
# Dev helper (GPU box): kernel trace of a pixel-starved shard (C5's 1/8 share, one mrt_redraw per frame): how many render kernels run side by side
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/trace_shard8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MRT_NOBATCH=1 MRT_WARMUP=10 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o trace -- python3 $REPO/scripts/shard_throughput.py stress 1920 1080 4096 0 8 12 0 > $O/run.log 2>&1
tail -n 2 $O/run.log | cut -c1-200
find $O -name "*kernel_trace.csv" | head -2
