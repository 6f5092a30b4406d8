#!/bin/bash
# Dev helper (GPU box): only the shard-throughput block of scripts/refresh_measurements.sh (same commands), into gpurun_out/final/
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/final; mkdir -p $O
( MRT_WARMUP=6 python scripts/shard_throughput.py stress 1920 1080 4096 0 1 4 0
  MRT_WARMUP=4 python scripts/shard_throughput.py stress 1920 1080 4096 0 1 3 1
  MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 32 0      # one mrt_redraw per frame: 8 frames in flight (32 frames: the timed frames start together on an empty chip, which costs the first eight their stagger)
  MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 5 8 32 0
  MRT_WARMUP=16 MRT_NOBATCH=1 MRT_SLOTS=2 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 6 0     # round 3's schedule: 2 frames in flight on all waves
  MRT_WARMUP=8 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 6 1
  MRT_WARMUP=32 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 4 24 0      # (enough frames for the controller's trials to be over and for a few convoys of frames)
  MRT_WARMUP=28 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 16 0
  MRT_WARMUP=8 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 6 0
  MRT_WARMUP=8 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 6 1 ) > $O/shard_throughput.txt 2>&1
