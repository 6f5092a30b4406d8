#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05m; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
MRT_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --verify > $O/forced.json 2> $O/forced.err; echo "forced rc=$?"
MRT_BENCH_ABI_DEVICES=0,0 MRT_BENCH_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline --verify --steps 2 --warmup 1 > $O/gloo2.json 2> $O/gloo2.err; echo "gloo2 rc=$?"
python - <<'PY'
import json
for n in ("bench", "forced", "gloo2"):
    d = json.load(open(f"gpurun_out/r05m/{n}.json"))
    print(n, round(d["value"]), d["schedule"], d.get("gathered_image_equals_unsharded_frame"), {k: (round(v["value"]) if isinstance(v, dict) and "value" in v else v) for k, v in (d.get("other_configs") or {}).items() if k not in ("note",)})
PY
