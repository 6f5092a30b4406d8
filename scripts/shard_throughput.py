#!/usr/bin/env python3
"""Dev helper (GPU box): pipelined throughput and lane utilisation of ONE rank's share of a multi-GPU workload.
   scripts/shard_throughput.py <scene: cover-glass|stress> <w> <h> <spp> <rank> <world> [frames] [rng_mode 0|1]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M
scene = sys.argv[1]
w, h, spp, rank, world = (int(x) for x in sys.argv[2:7])
K = int(sys.argv[7]) if len(sys.argv) > 7 else 4
mode = int(sys.argv[8]) if len(sys.argv) > 8 else 0
sp, cam = M.scene_stress(1, 100) if scene == "stress" else M.scene_cover(1, True)
with M.State(M.Args(w, h, spp, 50, 1.0), seed=1, shard=(rank, world) if world > 1 else None) as st:
    from myraytracer_amd import _lib
    if os.environ.get("MRT_SWEEP"):       # 1 = SGPR-fed VALU sweep, 2 = matrix-core sweep
        assert _lib.load().mrt_debug_set_sweep(st._ctx, int(os.environ["MRT_SWEEP"])) == 0
    if os.environ.get("MRT_HIER"):        # "max_levels,top_target"
        h_ = [int(x) for x in os.environ["MRT_HIER"].split(",")]
        assert _lib.load().mrt_debug_set_hierarchy(st._ctx, h_[0], h_[1]) == 0
    if os.environ.get("MRT_WAVES_PER_CU"):   # persistent waves per CU (0 = what the kernel's registers / LDS admit)
        assert _lib.load().mrt_debug_set_schedule(st._ctx, 1, int(os.environ["MRT_WAVES_PER_CU"])) == 0
    if os.environ.get("MRT_SLOTS"):          # frames in flight
        assert _lib.load().mrt_debug_set_frames_in_flight(st._ctx, int(os.environ["MRT_SLOTS"])) == 0
    if os.environ.get("MRT_HINT"):           # "div,mult": pin the launch schedule
        st.set_schedule_hint(*[int(x) for x in os.environ["MRT_HINT"].split(",")])
    if os.environ.get("MRT_NOBATCH"):     # mrt_render without sharing launches among frames
        assert _lib.load().mrt_debug_set_frame_batching(st._ctx, 0) == 0
    if os.environ.get("MRT_CLUSTER"):
        _lib.load().mrt_debug_set_cluster_factor(st._ctx, float(os.environ["MRT_CLUSTER"]))
    st.set_world(sp); st.set_camera(cam); st.set_rng_mode(mode)
    st.render(int(os.environ.get('MRT_WARMUP', '8'))); st.sync()       # (a pixel-starved shard runs up to 8 frames at a time)
    c0 = st.read_counters()
    t0 = time.perf_counter(); st.render(K); st.sync(); dt = time.perf_counter() - t0
    c1 = st.read_counters()
    util = (c1["world_hit_calls"] - c0["world_hit_calls"]) / max(1, c1["lane_slots"] - c0["lane_slots"])
    print(f"{scene} {w}x{h}x{spp} shard {rank}/{world} rng_mode {mode}: {dt / K * 1e3:.1f} ms/frame, per-GPU "
          f"{w * h * spp / world * K / dt * 1e-6:.1f} Msamples/s, lane utilisation {util:.3f}, "
          f"kernel ms {[round(x) for x in st.kernel_ms_history(K)]}, schedule {st.get_schedule()}", flush=True)
