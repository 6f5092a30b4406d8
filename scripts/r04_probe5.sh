#!/bin/bash
# Dev helper (GPU box), round 4: automatic frames-in-flight policy (warm), reject-cap A/B on C3 and C5, bench c5
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe5; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden_and_api.py tests/test_gpu_random_scenes.py -m gpu -q -x 2>&1 | tail -3 | tee $O/tests.txt
for cap in 0 3 4 5 6; do
  MRT_REJECT_CAP=$cap python scripts/wall_rate.py cover-glass 1920 1080 512 8 2>/dev/null | sed "s/^/cap=$cap /" | tee -a $O/cap.txt
done
for cap in 0 4; do
  MRT_REJECT_CAP=$cap python scripts/wall_rate.py stress 1920 1080 512 4 2>/dev/null | sed "s/^/cap=$cap /" | tee -a $O/cap.txt
  MRT_REJECT_CAP=$cap python scripts/wall_rate.py default 400 225 16 200 2>/dev/null | sed "s/^/cap=$cap /" | tee -a $O/cap.txt
done
( MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 16 0
  MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 4 8 0
  MRT_SLOTS=2 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 4 8 0
  MRT_SLOTS=4 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 4 8 0
  MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 4 0
  MRT_SLOTS=4 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 8 0
  ) 2>/dev/null | tee $O/shards.txt
python bench.py --config c5 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 400 $O/bench_c5.err; head -c 300 $O/bench_c5.json
