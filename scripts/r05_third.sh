#!/bin/bash
# Dev helper (GPU box): C5's 1/8 share under different schedules (frames in flight x launch width), incl. over-subscription
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r05c; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_schedule.py -m gpu -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" | tee -a $O/tests.txt
tail -n 5 $O/tests.txt
W="timeout -k 10 200 python scripts/wall_rate.py stress 1920 1080 4096 32"
( for H in 8,1 4,2 2,4 4,1; do MRT_HINT=$H MRT_WARMUP=24 MRT_SHARD=0,8 $W; done
  export MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_slots16.so
  for Q in 16 32; do GPU_MAX_HW_QUEUES=$Q python -c "
import myraytracer_amd as M
with M.State(M.Args(64, 40, 2, 8, 1.0), seed=1) as st:
    print('GPU_MAX_HW_QUEUES=$Q: streams running side by side of 8 / 12 / 16:', st.debug_stream_concurrency(8), st.debug_stream_concurrency(12), st.debug_stream_concurrency(16))
"; done
  for H in 8,2 4,4 4,3; do GPU_MAX_HW_QUEUES=32 MRT_HINT=$H MRT_WARMUP=32 MRT_SHARD=0,8 $W; done ) > $O/wall.txt 2>&1
cat $O/wall.txt
MRT_HINT=8,1 MRT_WARMUP=32 MRT_WAVE_SLOTS=4096 MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so timeout -k 10 300 python scripts/shard_occupancy.py stress 1920 1080 4096 0 8 32 > $O/shard_occupancy_8x1.txt 2>&1; echo "occupancy rc=$?"
head -n 3 $O/shard_occupancy_8x1.txt
