#!/usr/bin/env python3
"""After scripts/refresh_measurements.sh (gpurun_out/final/): copy what is judged into profiles/ under the round's prefix.
   python scripts/collect_profiles.py r03"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
src, dst = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles")
names = {"bench_n1.json": "bench_n1.json", "bench_c4.json": "bench_c4_n1.json", "bench_c5.json": "bench_c5_n1.json",
         "bench_c5_counter.json": "bench_c5_counter_n1.json", "bench_c1.json": "bench_c1_n1.json", "bench_c1_x32.json": "bench_c1_x32_n1.json",
         "bench_c2.json": "bench_c2_n1.json", "bench_c2_x32.json": "bench_c2_x32_n1.json", "bench_interactive.json": "bench_interactive_n1.json",
         "bench_interactive_x32.json": "bench_interactive_x32_n1.json", "bench_forced_dist.json": "bench_forced_dist_n1.json",
         "rehearsal_gloo_n2.json": "rehearsal_gloo_n2_one_gpu.json", "config_rates.txt": "config_rates.txt",
         "shard_throughput.txt": "shard_throughput.txt", "c3_phase.txt": "c3_phase_profile.txt", "c5_phase.txt": "c5_phase_profile.txt",
         "interactive_phase.txt": "interactive_phase_profile.txt", "bench_c2_again.json": "bench_c2_again_n1.json",
         "bench_c2_measure.json": "bench_c2_measure_n1.json", "bench_n1_measure.json": "bench_measure_n1.json",
         "shard_occupancy.txt": "shard_occupancy.txt", "controller_outcomes.txt": "controller_outcomes.txt", "viewer_rates.txt": "viewer_rates.txt"}
for a, b in names.items():
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{rnd}_{b}"))
    else:
        print("missing:", a)
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "summarize_profile.py"), "c3", rnd, "cover-glass_1920x1080x512_n1"], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "summarize_profile.py"), "c5", rnd, "stress_1920x1080x4096_n1"], stdout=subprocess.DEVNULL)
print(open(os.path.join(src, "gputests.log")).read().strip().splitlines()[-1])
