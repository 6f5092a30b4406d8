#!/bin/bash
# Dev helper (GPU box): bench.py --config c1 a few times with the launch-width controller's decisions traced
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$1; mkdir -p $O
for r in 1 2 3 4 5; do
  MRT_TRACE_WIDTH=1 python bench.py --config c1 --no-cpu-baseline --steps 800 --warmup 600 2> $O/err_$r.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench c1', round(d['value']), d['ms_per_step'], d['valu']['lane_utilisation'])"
  grep "mrt width" $O/err_$r.txt | cut -c1-140
done | tee $O/c1.txt
