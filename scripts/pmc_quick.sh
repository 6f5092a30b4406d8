#!/bin/bash
# Dev helper (GPU box): two rocprofv3 --pmc passes over scripts/quick_counters.py; prints per-kernel sums.
#   scripts/pmc_quick.sh <tag> [quick_counters args...]
set -e
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/p1 -o p1 -- python3 $REPO/scripts/quick_counters.py "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o p2 -- python3 $REPO/scripts/quick_counters.py "$@" > $OUT/p2.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if "render_kernel" not in k: continue
        print(k)
        for c, v in sorted(d.items()): print(f"   {c:28s} {v:.4e}  ({n[(k, c)]} dispatches)")
PY
tail -2 $OUT/p1.log
