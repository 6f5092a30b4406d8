#!/bin/bash
# Dev helper (GPU box): C5 at 512 spp, wall-clock rate of 6 pipelined frames, under different hierarchy depths / top sizes
for h in "3,256" "4,64" "3,256" "4,64" "4,16" "2,700"; do
  MRT_HIER=$h python scripts/wall_rate.py stress 1920 1080 512 6 2>/dev/null
done
MRT_BOXES=0 python scripts/wall_rate.py stress 1920 1080 512 6 2>/dev/null
MRT_RNG=1 python scripts/wall_rate.py stress 1920 1080 512 6 2>/dev/null
MRT_RNG=1 MRT_HIER=4,64 python scripts/wall_rate.py stress 1920 1080 512 6 2>/dev/null
