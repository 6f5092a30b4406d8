#!/bin/bash
# Dev helper (GPU box): texture-path counters of the render kernel at C5 and C3 (separate rocprofv3 --pmc passes, no tracing)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
O=$REPO/gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -oE "\b(TA_[A-Z_]+|TCP_[A-Z_0-9]+|TD_[A-Z_]+)\b" | sort -u > $O/avail.txt
wc -l $O/avail.txt
for cfg in c5 c3; do
  if [ $cfg = c5 ]; then ARGS="$REPO/bench.py --no-cpu-baseline --config c5 --steps 2 --warmup 2"; else ARGS="$REPO/bench.py --no-cpu-baseline --steps 4 --warmup 2"; fi
  rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE --output-format csv -d $O/ta_$cfg -o ta -- python3 $ARGS > $O/ta_$cfg.log 2>&1 || echo "pass 1 failed for $cfg"
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $O/tcp_$cfg -o tcp -- python3 $ARGS > $O/tcp_$cfg.log 2>&1 || echo "pass 2 failed for $cfg"
done
find $O -name "*counter_collection.csv" | head
