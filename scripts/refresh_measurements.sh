#!/bin/bash
# Everything profiles/ holds for a round, in one gpurun call (run on the GPU box from the repo root):
#   gpurun --timeout 1190 -- 'bash scripts/refresh_measurements.sh a'     benches, rocprofv3 passes
#   gpurun --timeout 1190 -- 'bash scripts/refresh_measurements.sh b'     shares, occupancy, controller, phases (a call is 20 minutes at most)
# Writes under gpurun_out/final/; afterwards, here:  python scripts/collect_profiles.py r04   copies what is judged into profiles/.
# Needs `make all stamps` beforehand.  About 15 GPU-minutes.  (bench.py pins the schedules of profiles/schedules.json and warms
# up until three generations of frames have gone through the pipeline, so the warm-up counts here are small.)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/final; mkdir -p $O
PART=${1:-a}
if [ "$PART" = "a" ]; then
python -m pytest tests -m gpu -q > $O/gputests.log 2>&1 || true; tail -2 $O/gputests.log
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
python bench.py --config c4 --steps 16 --warmup 2 > $O/bench_c4.json 2>/dev/null
python bench.py --config c5 --steps 16 --warmup 4 > $O/bench_c5.json 2>/dev/null
python bench.py --config c5 --rng counter --no-cpu-baseline --steps 12 --warmup 2 > $O/bench_c5_counter.json 2>/dev/null
python bench.py --config c1 --steps 3000 --warmup 5 > $O/bench_c1.json 2>/dev/null
python bench.py --config c1 --no-cpu-baseline --frames-per-step 32 --steps 20 --warmup 2 > $O/bench_c1_x32.json 2>/dev/null
python bench.py --config c2 --steps 150 --warmup 5 > $O/bench_c2.json 2>/dev/null
python bench.py --config c2 --no-cpu-baseline --steps 150 --warmup 5 > $O/bench_c2_again.json 2>/dev/null          # the same command again: a pinned schedule reproduces
python bench.py --config c2 --no-cpu-baseline --steps 150 --warmup 5 --schedule measure > $O/bench_c2_measure.json 2>/dev/null
python bench.py --no-cpu-baseline --no-other-configs --steps 20 --warmup 5 --schedule measure > $O/bench_n1_measure.json 2>/dev/null
python bench.py --config c2 --no-cpu-baseline --frames-per-step 32 --steps 4 --warmup 1 > $O/bench_c2_x32.json 2>/dev/null
python bench.py --config interactive --no-cpu-baseline --steps 3000 --warmup 5 > $O/bench_interactive.json 2>/dev/null
python bench.py --config interactive --no-cpu-baseline --frames-per-step 32 --steps 20 --warmup 2 > $O/bench_interactive_x32.json 2>/dev/null
MRT_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --verify > $O/bench_forced_dist.json 2>/dev/null
MRT_BENCH_ABI_DEVICES=0,0 MRT_BENCH_BACKEND=gloo python bench.py --gpus 2 --no-cpu-baseline --verify --steps 2 --warmup 1 > $O/rehearsal_gloo_n2.json 2>/dev/null
echo benches done
bash scripts/profile.sh c3 --steps 4 --warmup 2 --hint 1,1 > $O/profile_c3.log 2>&1      # (full-width launches: the profiler runs them one at a time)
bash scripts/profile.sh c5 --config c5 --steps 2 --warmup 2 > $O/profile_c5.log 2>&1
echo profiles done
exit 0
fi
python scripts/config_rates.py > $O/config_rates.txt 2>&1
# one mrt_redraw per frame (MRT_NOBATCH), the library's own schedule unless a line says otherwise
( MRT_WARMUP=12 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 1 8 0
  MRT_WARMUP=4 python scripts/shard_throughput.py stress 1920 1080 4096 0 1 3 1
  MRT_WARMUP=32 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 48 0      # 16 frames in flight on an eighth of the waves each
  MRT_WARMUP=32 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 5 8 48 0
  MRT_WARMUP=32 MRT_NOBATCH=1 MRT_HINT=8,1 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 48 0     # round 4's schedule: 8 in flight, one per eighth
  MRT_WARMUP=16 MRT_NOBATCH=1 MRT_SLOTS=2 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 6 0     # round 3's schedule: 2 frames in flight on all waves
  MRT_WARMUP=8 python scripts/shard_throughput.py stress 1920 1080 4096 0 8 6 1
  MRT_WARMUP=32 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 4 32 0
  MRT_WARMUP=24 MRT_NOBATCH=1 python scripts/shard_throughput.py stress 1920 1080 4096 0 2 16 0
  MRT_WARMUP=16 MRT_NOBATCH=1 MRT_HINT=4,2 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 32 0     # (profiles/schedules.json: c4_n8)
  MRT_WARMUP=8 MRT_NOBATCH=1 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 16 0
  MRT_WARMUP=8 python scripts/shard_throughput.py cover-glass 3840 2160 1024 0 8 6 1 ) > $O/shard_throughput.txt 2>&1
# where the wave slots of C5's 1/8 share go: round 4's schedule, then this round's
( MRT_HINT=8,1 MRT_WARMUP=32 MRT_WAVE_SLOTS=4096 MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/shard_occupancy.py stress 1920 1080 4096 0 8 32
  MRT_HINT=8,2 MRT_WARMUP=48 MRT_WAVE_SLOTS=4096 MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/shard_occupancy.py stress 1920 1080 4096 0 8 32 ) > $O/shard_occupancy.txt 2>&1
# what the library's controller arrives at by itself (every decision: MRT_TRACE_WIDTH), two fresh starts each
MRT_TRACE_WIDTH=1 MRT_ONLY=c1_n1,c2_n1,c3_n1,c4_n8,c5_n8,interactive_n1 python scripts/settle_schedules.py $O/controller_outcomes.json 2 > $O/controller_outcomes.txt 2>&1
# a viewer that reads every frame back (never more than one frame in flight): C5's 1/8 share and 1080p at 1 spp
( MRT_WARMUP=20 MRT_SHARD=0,8 MRT_READ_EVERY=1 python scripts/wall_rate.py stress 1920 1080 4096 6
  MRT_WARMUP=40 MRT_READ_EVERY=1 python scripts/wall_rate.py cover-glass 1920 1080 1 200 ) > $O/viewer_rates.txt 2>&1
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py cover-glass 1920 1080 64 > $O/c3_phase.txt
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py stress 1920 1080 64 > $O/c5_phase.txt
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py cover-glass 1920 1080 1 32 > $O/interactive_phase.txt
echo all done
