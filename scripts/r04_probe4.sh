#!/bin/bash
# Dev helper (GPU box), round 4: the automatic frames-in-flight policy of pixel-starved shards, hardware queues set by the package
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe4; mkdir -p $O
( MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 16 0
  MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 3 8 16 0
  GPU_MAX_HW_QUEUES=4 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 16 0
  MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 4 8 0
  MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 4 0
  MRT_NOBATCH=1 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 6 0 ) 2>/dev/null | tee $O/shards.txt
python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench c5', d['value'], d['ms_per_step'], d['valu']['lane_utilisation'])" | tee -a $O/shards.txt
timeout -k 10 600 python -m pytest tests/test_gpu_fullspp.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/shards.txt
