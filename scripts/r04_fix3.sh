#!/bin/bash
# Dev helper (GPU box): parity tests (all), then C3 / C2 / C4 / C1 / C5 rates
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1 || { tail -20 $O/tests.log; exit 1; }
tail -2 $O/tests.log
( for r in 1 2 3; do MRT_WARMUP=4 python scripts/wall_rate.py cover-glass 1920 1080 512 12; done
  MRT_WARMUP=60 python scripts/wall_rate.py cover 1200 675 64 60
  MRT_WARMUP=4 python scripts/wall_rate.py cover-glass 3840 2160 1024 3
  MRT_WARMUP=6 python scripts/wall_rate.py stress 1920 1080 512 10 ) 2>/dev/null | tee $O/rates.txt
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py cover-glass 1920 1080 64 > $O/c3_phase.txt 2>/dev/null
cat $O/c3_phase.txt
