#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05k; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_schedule.py -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests.txt
tail -n 5 $O/tests.txt
run() { MRT_SHARD=$1 MRT_HINT=$2 MRT_WARMUP=$3 timeout -k 10 300 python scripts/wall_rate.py $5 $6 $7 $8 $4 2>&1 | grep -v amdgpu.ids | sed -e 's/HIER=None BOXES=None RNG=None//' -e "s/^/shard $1 hint $2: /" | cut -c1-120; }
( for H in 1,1 4,2 2,2 1,1 4,2 2,2; do run 0,8 $H 16 32 cover-glass 3840 2160 1024; done
  MRT_WARMUP=20 MRT_SHARD=0,8 MRT_READ_EVERY=1 python scripts/wall_rate.py stress 1920 1080 4096 6 2>&1 | grep -v amdgpu | cut -c1-330 ) > $O/rates.txt 2>&1
cat $O/rates.txt
