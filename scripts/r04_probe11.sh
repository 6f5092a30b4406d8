#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe11; mkdir -p $O
( MRT_WARMUP=12 python scripts/wall_rate.py cover-glass 1920 1080 512 16
  MRT_WARMUP=6 python scripts/wall_rate.py cover-glass 3840 2160 1024 4
  MRT_WARMUP=60 python scripts/wall_rate.py cover 1200 675 64 60
  MRT_WARMUP=60 python scripts/wall_rate.py default 400 225 16 400
  MRT_WARMUP=10 python scripts/wall_rate.py stress 1920 1080 512 8
  MRT_WARMUP=6 python scripts/wall_rate.py stress 1920 1080 4096 4
  for n in 36 50 70; do MRT_WARMUP=30 python scripts/wall_rate.py stress$n 1920 1080 64 16; done
  MRT_WARMUP=40 python scripts/wall_rate.py cover-glass 1920 1080 8 100
  MRT_WARMUP=40 python scripts/wall_rate.py cover-glass 1920 1080 4 200
  MRT_WARMUP=40 python scripts/wall_rate.py cover-glass 1920 1080 1 400
  ) 2>/dev/null | tee $O/rates.txt
( MRT_WARMUP=12 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 8 0
  MRT_WARMUP=12 MRT_NOBATCH=1 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 8 0
  MRT_WARMUP=24 MRT_NOBATCH=1 python scripts/shard_throughput.py cover-glass 1920 1080 512 0 8 16 0 ) 2>/dev/null | tee $O/shards.txt
python bench.py --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench', d['value'], d['ms_per_step'])"
python bench.py --config c2 --no-cpu-baseline --steps 60 --warmup 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench c2', d['value'], d['ms_per_step'], d['valu']['lane_utilisation'])"
python bench.py --config c1 --no-cpu-baseline --steps 400 --warmup 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench c1', d['value'], d['ms_per_step'], d['valu']['lane_utilisation'])"
