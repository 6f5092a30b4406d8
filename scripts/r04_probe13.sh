#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe13; mkdir -p $O
( MRT_WARMUP=8 MRT_NOBATCH=1 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 12 0
  MRT_WARMUP=8 MRT_NOBATCH=1 MRT_SLOTS=4 MRT_WAVES_PER_CU=5 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 12 0
  MRT_WARMUP=8 MRT_NOBATCH=1 MRT_SLOTS=2 MRT_WAVES_PER_CU=10 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 12 0
  MRT_WARMUP=8 MRT_NOBATCH=1 MRT_SLOTS=8 MRT_WAVES_PER_CU=3 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 16 0
  MRT_WARMUP=8 MRT_NOBATCH=1 python scripts/shard_throughput.py cover-glass 2716 1528 512 0 2 12 0
  MRT_WARMUP=8 MRT_NOBATCH=1 python scripts/shard_throughput.py cover-glass 3840 2160 512 0 4 12 0
  ) 2>/dev/null | tee $O/shards.txt
MRT_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --verify > $O/forced.json 2> $O/forced.err; python -c "
import json; d=json.load(open('$O/forced.json')); print('forced dist', d['value'], d.get('gathered_image_equals_unsharded_frame'), d.get('abi_rccl_gather'))"
MRT_BENCH_ABI_DEVICES=0,0 MRT_BENCH_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline --verify --steps 2 --warmup 1 > $O/gloo2.json 2> $O/gloo2.err; python -c "
import json; d=json.load(open('$O/gloo2.json')); print('gloo2', d['value'], d.get('gathered_image_equals_unsharded_frame'), d.get('abi_single_process'), [r.get('ms_per_step') for r in d['ranks']])"
MRT_BENCH_ABI_DEVICES=0,0 MRT_BENCH_BACKEND=gloo python bench.py --gpus 2 --config c5 --no-cpu-baseline --verify --steps 2 --warmup 1 --no-abi-legs > $O/gloo2c5.json 2> $O/gloo2c5.err; python -c "
import json; d=json.load(open('$O/gloo2c5.json')); print('gloo2 c5', d['value'], d.get('gathered_image_equals_unsharded_frame'), d['config']['workload'][:200])"
tail -3 $O/*.err | cut -c1-300
