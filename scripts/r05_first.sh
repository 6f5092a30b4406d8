#!/bin/bash
# Dev helper (GPU box): round 5's first pass -- the GPU tests, the headline, and where the wave slots of C5's 1/8 share go.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05a; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests.txt
tail -n 3 $O/tests.txt
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cut -c1-400 $O/bench.json
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so timeout -k 10 300 python scripts/shard_occupancy.py stress 1920 1080 4096 0 8 24 > $O/shard_occupancy.txt 2>&1; echo "occupancy rc=$?"
head -n 12 $O/shard_occupancy.txt
( MRT_WARMUP=24 MRT_SHARD=0,8 timeout -k 10 200 python scripts/wall_rate.py stress 1920 1080 4096 24
  MRT_WARMUP=4 MRT_SHARD=0,8 MRT_READ_EVERY=1 timeout -k 10 200 python scripts/wall_rate.py stress 1920 1080 4096 6
  MRT_WARMUP=8 MRT_READ_EVERY=1 timeout -k 10 200 python scripts/wall_rate.py cover-glass 1920 1080 1 200
  MRT_WARMUP=8 timeout -k 10 200 python scripts/wall_rate.py cover-glass 1920 1080 1 200 ) > $O/wall.txt 2>&1
cat $O/wall.txt
