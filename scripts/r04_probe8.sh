#!/bin/bash
# Dev helper (GPU box), round 4: frames in flight x waves per launch for frames that are NOT pixel-starved
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe8; mkdir -p $O
( MRT_SLOTS=4 MRT_SCHED=2,4 python scripts/wall_rate.py stress 1920 1080 4096 6
  MRT_SLOTS=8 MRT_SCHED=2,2 python scripts/wall_rate.py stress 1920 1080 4096 8
  MRT_SLOTS=2 MRT_SCHED=2,8 python scripts/wall_rate.py stress 1920 1080 4096 4
  python scripts/wall_rate.py cover-glass 1920 1080 512 16
  MRT_SLOTS=4 MRT_SCHED=2,5 python scripts/wall_rate.py cover-glass 1920 1080 512 16
  MRT_SLOTS=8 MRT_SCHED=2,3 python scripts/wall_rate.py cover-glass 1920 1080 512 16
  MRT_SLOTS=4 MRT_SCHED=2,10 python scripts/wall_rate.py cover-glass 1920 1080 512 16
  MRT_SLOTS=2 MRT_SCHED=2,10 python scripts/wall_rate.py cover-glass 1920 1080 512 16
  python scripts/wall_rate.py cover-glass 3840 2160 1024 4
  MRT_SLOTS=4 MRT_SCHED=2,5 python scripts/wall_rate.py cover-glass 3840 2160 1024 8
  ) 2>/dev/null | tee $O/slots.txt
