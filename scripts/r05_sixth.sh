#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05g; mkdir -p $O
( for Q in 18 20 18 20 18 20 32 32; do echo -n "GPU_MAX_HW_QUEUES=$Q: "; GPU_MAX_HW_QUEUES=$Q MRT_SHARD=0,8 MRT_HINT=8,2 MRT_WARMUP=32 timeout -k 10 200 python scripts/wall_rate.py stress 1920 1080 4096 48 2>&1 | grep -v amdgpu | cut -c52-110; done ) | tee $O/queues3.txt
