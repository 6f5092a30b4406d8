#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe19; mkdir -p $O
( MRT_WARMUP=300 python scripts/wall_rate.py cover-glass 1920 1080 1 800
  MRT_WARMUP=300 python scripts/wall_rate.py cover-glass 1920 1080 2 400
  MRT_WARMUP=300 python scripts/wall_rate.py cover-glass 1920 1080 3 400
  MRT_WARMUP=300 python scripts/wall_rate.py default 400 225 1 2000
  MRT_WARMUP=300 python scripts/wall_rate.py stress 1920 1080 1 400
  MRT_WARMUP=200 python scripts/wall_rate.py cover-glass 1920 1080 4 200
  MRT_WARMUP=200 python scripts/wall_rate.py default 400 225 16 800 ) 2>/dev/null | tee $O/rates.txt
python bench.py --config interactive --no-cpu-baseline --steps 800 --warmup 300 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench interactive', d['value'], d['ms_per_step'], d['valu']['lane_utilisation'])"
timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_golden_and_api.py tests/test_gpu_parity.py -m gpu -q -x > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt | cut -c1-300
