#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe7; mkdir -p $O
L=$PWD/myraytracer_amd/lib
for cap in 0 3 4 5; do
  MRT_REJECT_CAP=$cap python scripts/wall_rate.py cover-glass 1920 1080 512 8 2>/dev/null | sed "s/^/cap=$cap /" | tee -a $O/ab.txt
done
MRT_LIB_OVERRIDE=$L/libmrt_oldrej.so python scripts/wall_rate.py cover-glass 1920 1080 512 8 2>/dev/null | sed "s/^/oldrej /" | tee -a $O/ab.txt
for cap in 0 4; do
  MRT_REJECT_CAP=$cap python scripts/wall_rate.py stress 1920 1080 512 4 2>/dev/null | sed "s/^/cap=$cap /" | tee -a $O/ab.txt
  MRT_REJECT_CAP=$cap python scripts/wall_rate.py cover 1200 675 64 40 2>/dev/null | sed "s/^/cap=$cap /" | tee -a $O/ab.txt
done
timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -q -x > $O/tests.txt 2>&1; tail -5 $O/tests.txt | cut -c1-300
