#!/bin/bash
# Dev helper (GPU box): the GPU tests, the controller's outcomes (trace) and bench.py on the current build
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05i; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests.txt
tail -n 4 $O/tests.txt
MRT_TRACE_WIDTH=1 MRT_ONLY=c1_n1,c2_n1,c3_n1,c4_n8,interactive_n1 timeout -k 10 600 python scripts/settle_schedules.py $O/schedules_a.json 2 > $O/settle_a.txt 2>&1
grep -v "amdgpu\|mrt width: frame" $O/settle_a.txt | cut -c1-200
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --schedule measure --no-other-configs --no-cpu-baseline > $O/bench_measure.json 2> $O/bench_measure.err; echo "bench(measure) rc=$?"
python - <<'PY'
import json
for name in ("bench", "bench_measure"):
    d = json.load(open(f"gpurun_out/r05i/{name}.json"))
    print(name, round(d["value"]), round(d["ms_per_step"], 2), d["schedule"])
    for k, v in (d.get("other_configs") or {}).items():
        if isinstance(v, dict): print("  ", k, {kk: (round(v[kk], 1) if isinstance(v[kk], float) else v[kk]) for kk in ("value", "ms_per_step", "settled", "lane_utilisation", "error") if kk in v})
PY
