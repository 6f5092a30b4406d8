#!/bin/bash
# Dev helper (GPU box): the GPU tests on the 16-slot build, the schedules every workload settles at, bench.py with them
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05d; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests.txt
tail -n 4 $O/tests.txt
timeout -k 10 900 python scripts/settle_schedules.py $O/schedules.json 2 > $O/settle.txt 2>&1; echo "settle rc=$?"
grep -v amdgpu.ids $O/settle.txt | cut -c1-330
cp $O/schedules.json profiles/schedules.json
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05d/bench.json"))
print(d["value"], d["ms_per_step"], d["schedule"])
print(json.dumps(d.get("other_configs"), indent=1)[:3000])
PY
