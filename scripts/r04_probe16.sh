#!/bin/bash
# Dev helper (GPU box), round 4: more frames in flight for frames too small or too short to fill the chip
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe16; mkdir -p $O
( for s in 0 3 4 6 8; do MRT_SLOTS=$s MRT_WARMUP=40 python scripts/wall_rate.py default 400 225 16 800 | sed "s/^/slots=$s /"; done
  for s in 0 3 4 8; do MRT_SLOTS=$s MRT_WARMUP=40 python scripts/wall_rate.py cover-glass 1920 1080 1 800 | sed "s/^/slots=$s /"; done
  for s in 0 4 8; do MRT_SLOTS=$s MRT_WARMUP=40 python scripts/wall_rate.py cover-glass 1920 1080 2 400 | sed "s/^/slots=$s /"; done
  for s in 0 4; do MRT_SLOTS=$s MRT_WARMUP=40 python scripts/wall_rate.py cover-glass 640 360 16 400 | sed "s/^/slots=$s /"; done
  for s in 0 4; do MRT_SLOTS=$s MRT_WARMUP=40 python scripts/wall_rate.py cover 1200 675 4 400 | sed "s/^/slots=$s /"; done
) 2>/dev/null | tee $O/rates.txt
