#!/bin/bash
# Dev helper (GPU box), round 4: the launch-width controller across configs (automatic), after warm-up
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe10; mkdir -p $O
( MRT_WARMUP=12 python scripts/wall_rate.py cover-glass 1920 1080 512 16
  MRT_WARMUP=6 python scripts/wall_rate.py cover-glass 3840 2160 1024 4
  MRT_WARMUP=40 python scripts/wall_rate.py cover 1200 675 64 60
  MRT_WARMUP=60 python scripts/wall_rate.py default 400 225 16 400
  MRT_WARMUP=10 python scripts/wall_rate.py stress 1920 1080 512 8
  MRT_WARMUP=6 python scripts/wall_rate.py stress 1920 1080 4096 4
  MRT_WARMUP=6 MRT_RNG=1 python scripts/wall_rate.py stress 1920 1080 4096 4
  for n in 36 50 70; do MRT_WARMUP=30 python scripts/wall_rate.py stress$n 1920 1080 64 16; done
  MRT_WARMUP=40 python scripts/wall_rate.py cover-glass 1920 1080 8 100
  MRT_WARMUP=40 python scripts/wall_rate.py cover-glass 1920 1080 1 400
  ) 2>/dev/null | tee $O/rates.txt
( MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 16 0
  MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 4 8 0
  MRT_WARMUP=12 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 8 0
  MRT_WARMUP=12 MRT_NOBATCH=1 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 8 0
  MRT_WARMUP=16 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 14 0 ) 2>/dev/null | tee $O/shards.txt
timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; tail -4 $O/tests.txt | cut -c1-300
