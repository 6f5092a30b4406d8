#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe14; mkdir -p $O
timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_superset.py tests/test_gpu_random_scenes.py tests/test_gpu_parity.py -m gpu -q -x > $O/tests.txt 2>&1; tail -n 3 $O/tests.txt
( MRT_WARMUP=8 python scripts/wall_rate.py stress 1920 1080 512 8
  MRT_WARMUP=4 python scripts/wall_rate.py stress 1920 1080 4096 4
  for n in 36 50 70; do MRT_WARMUP=20 python scripts/wall_rate.py stress$n 1920 1080 64 16; done
  MRT_WARMUP=8 python scripts/wall_rate.py cover-glass 1920 1080 512 16 ) 2>/dev/null | tee $O/rates.txt
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py stress 1920 1080 64 2>/dev/null | tee $O/c5_phase.txt
