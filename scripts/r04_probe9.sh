#!/bin/bash
# Dev helper (GPU box), round 4: two co-resident half-width launches vs full-width launches, across configs
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe9; mkdir -p $O
( for rep in 1 2; do
  python scripts/wall_rate.py cover-glass 1920 1080 512 16
  MRT_SCHED=2,10 python scripts/wall_rate.py cover-glass 1920 1080 512 16
  done
  python scripts/wall_rate.py cover-glass 3840 2160 1024 4
  MRT_SCHED=2,10 python scripts/wall_rate.py cover-glass 3840 2160 1024 4
  python scripts/wall_rate.py cover 1200 675 64 60
  MRT_SCHED=2,10 python scripts/wall_rate.py cover 1200 675 64 60
  MRT_SLOTS=4 MRT_SCHED=2,5 python scripts/wall_rate.py cover 1200 675 64 60
  python scripts/wall_rate.py stress 1920 1080 512 8
  MRT_SCHED=2,8 python scripts/wall_rate.py stress 1920 1080 512 8
  MRT_SLOTS=4 MRT_SCHED=2,4 python scripts/wall_rate.py stress 1920 1080 512 8
  MRT_SLOTS=3 MRT_SCHED=2,8 python scripts/wall_rate.py stress 1920 1080 4096 6
  MRT_SLOTS=4 MRT_SCHED=2,8 python scripts/wall_rate.py stress 1920 1080 4096 6
  MRT_RNG=1 MRT_SCHED=2,8 python scripts/wall_rate.py stress 1920 1080 4096 4
  for n in 36 70; do python scripts/wall_rate.py stress$n 1920 1080 64 16; MRT_SCHED=2,8 python scripts/wall_rate.py stress$n 1920 1080 64 16; done
  ) 2>/dev/null | tee $O/slots.txt
