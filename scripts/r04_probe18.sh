#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe18; mkdir -p $O
( for w in 4 6 8 12 20; do MRT_SLOTS=4 MRT_SCHED=2,$w MRT_WARMUP=100 python scripts/wall_rate.py cover-glass 1920 1080 1 800 | sed "s/^/slots=4 wpc=$w /"; done
  for w in 4 8 12; do MRT_SLOTS=8 MRT_SCHED=2,$w MRT_WARMUP=100 python scripts/wall_rate.py cover-glass 1920 1080 1 800 | sed "s/^/slots=8 wpc=$w /"; done
  for w in 8 12 20; do MRT_SLOTS=4 MRT_SCHED=2,$w MRT_WARMUP=100 python scripts/wall_rate.py cover-glass 1920 1080 2 400 | sed "s/^/slots=4 wpc=$w /"; done
) 2>/dev/null | tee $O/rates.txt
