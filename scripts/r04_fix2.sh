#!/bin/bash
# Dev helper (GPU box): large-scene parity tests, then C5 / mid-size scenes with the LDS-held boxes off / top only / top + next level
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_superset.py tests/test_gpu_random_scenes.py tests/test_gpu_parity.py -m gpu -q -x > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for cap in 0 160 1024; do
( export MRT_EXP_BOX_LDS=$cap
  MRT_WARMUP=6 python scripts/wall_rate.py stress 1920 1080 512 10
  for n in 36 50 70; do python scripts/wall_rate.py stress$n 1920 1080 64 8; done ) 2>/dev/null | sed "s/^/lds=$cap /" | tee -a $O/rates.txt
done
MRT_WARMUP=4 python scripts/wall_rate.py stress 1920 1080 4096 3 2>/dev/null | tee -a $O/rates.txt
MRT_WARMUP=4 python scripts/wall_rate.py cover-glass 1920 1080 512 12 2>/dev/null | tee -a $O/rates.txt
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py stress 1920 1080 64 > $O/c5_phase.txt 2>/dev/null
cat $O/c5_phase.txt
