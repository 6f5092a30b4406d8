#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe17; mkdir -p $O
( MRT_WARMUP=200 python scripts/wall_rate.py default 400 225 16 800
  MRT_WARMUP=200 python scripts/wall_rate.py cover-glass 1920 1080 1 800
  MRT_WARMUP=200 python scripts/wall_rate.py cover-glass 1920 1080 2 400
  MRT_WARMUP=100 python scripts/wall_rate.py cover-glass 640 360 16 400
  MRT_WARMUP=100 python scripts/wall_rate.py cover 1200 675 4 400
  MRT_WARMUP=60 python scripts/wall_rate.py cover 1200 675 64 60
  MRT_WARMUP=12 python scripts/wall_rate.py cover-glass 1920 1080 512 16
  MRT_WARMUP=6 python scripts/wall_rate.py stress 1920 1080 512 8 ) 2>/dev/null | tee $O/rates.txt
python bench.py --config c1 --no-cpu-baseline --steps 800 --warmup 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench c1', d['value'], d['ms_per_step'], d['valu']['lane_utilisation'])"
python bench.py --config interactive --no-cpu-baseline --steps 800 --warmup 200 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench interactive', d['value'], d['ms_per_step'], d['valu']['lane_utilisation'])"
python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench c3', d['value'], d['ms_per_step'])"
timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt | cut -c1-300
