#!/bin/bash
# Dev helper (GPU box), round 4: which change slowed C3 (A/B builds), the crashing test, more frames-in-flight points
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe6; mkdir -p $O
L=$PWD/myraytracer_amd/lib
for v in "" libmrt_nofast.so libmrt_oldrej.so libmrt_both.so; do
  if [ -n "$v" ]; then export MRT_LIB_OVERRIDE=$L/$v; else unset MRT_LIB_OVERRIDE; fi
  MRT_REJECT_CAP=4 python scripts/wall_rate.py cover-glass 1920 1080 512 8 2>/dev/null | sed "s/^/lib=$v /" | tee -a $O/ab.txt
done
unset MRT_LIB_OVERRIDE
( MRT_SLOTS=4 python scripts/wall_rate.py stress 1920 1080 4096 4
  MRT_SLOTS=3 python scripts/wall_rate.py stress 1920 1080 4096 3
  MRT_SLOTS=8 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 8 0
  MRT_SLOTS=4 MRT_WAVES_PER_CU=4 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 8 0
  python scripts/shard_throughput.py stress 1920 1080 4096 0 8 16 0 ) 2>/dev/null | tee $O/shards.txt
timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_golden_and_api.py tests/test_gpu_parity.py tests/test_gpu_random_scenes.py -m gpu -q -x > $O/tests.txt 2>&1; tail -60 $O/tests.txt | cut -c1-300
