#!/usr/bin/env python3
"""Dev helper (GPU box): pipelined throughput of ONE rank's share of a multi-GPU bench workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M
w, h, spp, rank, world = (int(x) for x in sys.argv[1:6])
K = int(sys.argv[6]) if len(sys.argv) > 6 else 4
sp, cam = M.scene_cover(1, True)
with M.State(M.Args(w, h, spp, 50, 1.0), seed=1, shard=(rank, world) if world > 1 else None) as st:
    if os.environ.get("MRT_SWEEP"):       # 1 = SGPR-fed VALU sweep, 2 = matrix-core sweep
        from myraytracer_amd import _lib
        assert _lib.load().mrt_debug_set_sweep(st._ctx, int(os.environ["MRT_SWEEP"])) == 0
    if os.environ.get("MRT_HIER"):        # "max_levels,top_target"
        from myraytracer_amd import _lib
        h_ = [int(x) for x in os.environ["MRT_HIER"].split(",")]
        assert _lib.load().mrt_debug_set_hierarchy(st._ctx, h_[0], h_[1]) == 0
    if os.environ.get("MRT_CLUSTER"):
        from myraytracer_amd import _lib
        _lib.load().mrt_debug_set_cluster_factor(st._ctx, float(os.environ["MRT_CLUSTER"]))
    st.set_world(sp); st.set_camera(cam); st.render(2); st.sync()
    t0 = time.perf_counter(); st.render(K); st.sync(); dt = time.perf_counter() - t0
    print(f"{w}x{h}x{spp} shard {rank}/{world}: {dt / K * 1e3:.1f} ms/frame, per-GPU {w * h * spp / world * K / dt * 1e-6:.1f} Msamples/s, "
          f"kernel ms {[round(x) for x in st.kernel_ms_history(K)]}")
