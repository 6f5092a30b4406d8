#!/bin/bash
# Dev helper (GPU box): C5 parity rows + rates + phase profile in one go.  usage: scripts/c5_quick.sh <outdir>
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_gpu_fullspp.py tests/test_gpu_superset.py tests/test_gpu_random_scenes.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
python bench.py --config c5 --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_c5.json 2>/dev/null
python bench.py --config c5 --rng counter --no-cpu-baseline --steps 2 --warmup 1 > $O/bench_c5_counter.json 2>/dev/null
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py stress 1920 1080 64 > $O/c5_phase.txt
cat $O/c5_phase.txt
python - <<PY
import json
for f in ["bench_c5","bench_c5_counter"]:
    d=json.load(open("$O/%s.json"%f)); print(f, round(d["value"]), round(d["ms_per_step"],1), "util", round(d["valu"]["lane_utilisation"],3))
PY
