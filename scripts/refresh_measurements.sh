#!/bin/bash
# Everything profiles/ holds for a round, in one gpurun call (run on the GPU box from the repo root):
#   gpurun --timeout 1150 -- 'bash scripts/refresh_measurements.sh'
# Writes under gpurun_out/final/; afterwards, here:  python scripts/collect_profiles.py r04   copies what is judged into profiles/.
# Needs `make all stamps` beforehand.  About 12 GPU-minutes.  (Warm-up steps: the launch-width controller of redraw_frames
# settles within ~3 x the frames in flight.)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || true; tail -2 $O/gputests.log
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --config c4 --steps 4 --warmup 2 > $O/bench_c4.json 2>/dev/null
python bench.py --config c5 --steps 4 --warmup 4 > $O/bench_c5.json 2>/dev/null
python bench.py --config c5 --rng counter --no-cpu-baseline --steps 3 --warmup 2 > $O/bench_c5_counter.json 2>/dev/null
python bench.py --config c1 --steps 800 --warmup 600 > $O/bench_c1.json 2>/dev/null
python bench.py --config c1 --no-cpu-baseline --frames-per-step 32 --steps 20 --warmup 2 > $O/bench_c1_x32.json 2>/dev/null
python bench.py --config c2 --steps 80 --warmup 60 > $O/bench_c2.json 2>/dev/null
python bench.py --config c2 --no-cpu-baseline --frames-per-step 32 --steps 4 --warmup 1 > $O/bench_c2_x32.json 2>/dev/null
python bench.py --config interactive --no-cpu-baseline --steps 800 --warmup 600 > $O/bench_interactive.json 2>/dev/null
python bench.py --config interactive --no-cpu-baseline --frames-per-step 32 --steps 20 --warmup 2 > $O/bench_interactive_x32.json 2>/dev/null
MRT_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --verify > $O/bench_forced_dist.json 2>/dev/null
MRT_BENCH_ABI_DEVICES=0,0 MRT_BENCH_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline --verify --steps 2 --warmup 1 > $O/rehearsal_gloo_n2.json 2>/dev/null
echo benches done
bash scripts/profile.sh c3 --steps 4 --warmup 2 > $O/profile_c3.log 2>&1
bash scripts/profile.sh c5 --config c5 --steps 2 --warmup 2 > $O/profile_c5.log 2>&1
echo profiles done
python scripts/config_rates.py > $O/config_rates.txt 2>&1
( MRT_WARMUP=6 python scripts/shard_throughput.py stress 1920 1080 4096 0 1 4 0
  MRT_WARMUP=4 python scripts/shard_throughput.py stress 1920 1080 4096 0 1 3 1
  MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 32 0      # one mrt_redraw per frame: 8 frames in flight (32 frames: the timed frames start together on an empty chip, which costs the first eight their stagger)
  MRT_WARMUP=16 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 5 8 32 0
  MRT_WARMUP=16 MRT_NOBATCH=1 MRT_SLOTS=2 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 6 0     # round 3's schedule: 2 frames in flight on all waves
  MRT_WARMUP=8 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 6 1
  MRT_WARMUP=32 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 4 24 0      # (enough frames for the controller's trials to be over and for a few convoys of frames)
  MRT_WARMUP=28 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 16 0
  MRT_WARMUP=8 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 6 0
  MRT_WARMUP=8 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 6 1 ) > $O/shard_throughput.txt 2>&1
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py cover-glass 1920 1080 64 > $O/c3_phase.txt
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py stress 1920 1080 64 > $O/c5_phase.txt
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py cover-glass 1920 1080 1 32 > $O/interactive_phase.txt
echo all done
