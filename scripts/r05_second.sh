#!/bin/bash
# Dev helper (GPU box): the schedule tests again (probe by device stamps), then where the wave slots of C5's 1/8 share go
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05b; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_schedule.py -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests.txt
tail -n 5 $O/tests.txt
MRT_WAVE_SLOTS=4096 MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so timeout -k 10 300 python scripts/shard_occupancy.py stress 1920 1080 4096 0 8 32 > $O/shard_occupancy.txt 2>&1; echo "occupancy rc=$?"
head -n 40 $O/shard_occupancy.txt
( MRT_WARMUP=24 MRT_SHARD=0,8 timeout -k 10 200 python scripts/wall_rate.py stress 1920 1080 4096 32
  MRT_WARMUP=4 MRT_SHARD=0,8 MRT_READ_EVERY=1 timeout -k 10 200 python scripts/wall_rate.py stress 1920 1080 4096 6
  MRT_WARMUP=12 timeout -k 10 200 python scripts/wall_rate.py stress 1920 1080 4096 8 ) > $O/wall.txt 2>&1
cat $O/wall.txt
