#!/bin/bash
# Dev helper (GPU box), round 4: the large-scene walk -- parity tests that exercise it, C5 rates, phase profile, smaller large scenes,
# and C3 (small-scene kernel shares the source).   usage: scripts/r04_c5.sh <tag> [notests]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/c5_$1; mkdir -p $O
if [ "$2" != "notests" ]; then
  timeout -k 10 900 python -m pytest tests/test_gpu_superset.py tests/test_gpu_random_scenes.py tests/test_gpu_parity.py tests/test_gpu_fullspp.py -m gpu -q -x > $O/tests.log 2>&1
  tail -3 $O/tests.log
fi
python scripts/wall_rate.py stress 1920 1080 512 6 2>/dev/null | tee -a $O/rates.txt
python scripts/wall_rate.py stress 1920 1080 4096 2 2>/dev/null | tee -a $O/rates.txt
MRT_RNG=1 python scripts/wall_rate.py stress 1920 1080 4096 2 2>/dev/null | tee -a $O/rates.txt
for n in 36 50 70; do python scripts/wall_rate.py stress$n 1920 1080 64 8 2>/dev/null | tee -a $O/rates.txt; done
python scripts/wall_rate.py cover-glass 1920 1080 512 8 2>/dev/null | tee -a $O/rates.txt
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py stress 1920 1080 64 > $O/c5_phase.txt 2>/dev/null
cat $O/c5_phase.txt
( MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 16 0
  MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 3 8 16 0
  MRT_NOBATCH=1 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 6 0 ) 2>/dev/null | tee $O/shards.txt
( GPU_MAX_HW_QUEUES=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 16 0 ) 2>/dev/null | sed "s/^/hwq=16 /" | tee -a $O/shards.txt
