#!/bin/bash
# Dev helper (GPU box): the launch-width controller's decisions (MRT_TRACE_WIDTH) over repeated starts of the same workload
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$1; mkdir -p $O
for r in 1 2 3 4 5 6 7 8 9 10; do MRT_TRACE_WIDTH=1 MRT_WARMUP=2 python scripts/wall_rate.py cover-glass 1920 1080 512 14 2>&1 | grep -v "amdgpu.ids" | cut -c1-160; done | tee $O/c3.txt
( MRT_TRACE_WIDTH=1 MRT_WARMUP=60 python scripts/wall_rate.py cover 1200 675 64 60
  MRT_TRACE_WIDTH=1 MRT_WARMUP=200 python scripts/wall_rate.py default 400 225 16 800
  MRT_TRACE_WIDTH=1 MRT_WARMUP=200 python scripts/wall_rate.py cover-glass 1920 1080 1 800 ) 2>&1 | grep -v "amdgpu.ids" | cut -c1-160 | tee $O/others.txt
for r in 1 2 3; do python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench c3', round(d['value']), d['ms_per_step'], d['valu']['lane_utilisation'])"; done | tee $O/bench.txt
