#!/bin/bash
# Dev helper (GPU box), round 4 baseline probe: bench + clocks under load + pixel-starved shard sweep over resident waves per CU.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/probe1; mkdir -p $O
python bench.py --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_c3.json 2> $O/bench_c3.err &
BP=$!
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do sleep 2; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" >> $O/smi.txt; echo --- >> $O/smi.txt; done
wait $BP
tail -c 600 $O/bench_c3.json
for w in 0 12 8 6 4; do
  MRT_NOBATCH=1 MRT_WAVES_PER_CU=$w python scripts/shard_throughput.py stress 1920 1080 4096 0 8 3 0 2>/dev/null | sed "s/^/wpc=$w /" >> $O/shard_sweep.txt
done
cat $O/shard_sweep.txt
MRT_LIB_OVERRIDE=$PWD/myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py stress 1920 1080 64 > $O/c5_phase.txt 2>/dev/null
cat $O/c5_phase.txt
python scripts/wall_rate.py stress 1920 1080 512 4 2>/dev/null | tee $O/c5_wall512.txt
