#!/bin/bash
# rocprofv3 passes over one bench.py invocation (run on the GPU box through gpurun).
#   scripts/profile.sh <tag> [bench.py args...]
# Writes CSV summaries under gpurun_out/prof_<tag>/ ; copy what should be judged to profiles/.
# Kernel trace and each PMC set run as separate passes (never combined with tracing domains).
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --no-cpu-baseline --no-other-configs $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc1 -o pmc1 -- python3 $ARGS > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM --output-format csv -d $OUT/pmc2 -o pmc2 -- python3 $ARGS > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc3 -o pmc3 -- python3 $ARGS > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -o pmc4 -- python3 $ARGS > $OUT/bench_pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
