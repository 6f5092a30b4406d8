#!/usr/bin/env python3
"""Dev helper (GPU box): pipelined Msamples/s of BASELINE's small configs (C1, C2) and of 1-spp interactive frames."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M
cases = [("C1 default 400x225x16 depth 8", M.scene_default(), None, 400, 225, 16, 8, 200),
         ("C2 cover 1200x675x64 depth 50", *M.scene_cover(1, False), 1200, 675, 64, 50, 40),
         ("cover-glass 1920x1080x1 depth 50 (interactive)", *M.scene_cover(1, True), 1920, 1080, 1, 50, 200),
         ("cover-glass 1920x1080x8 depth 50", *M.scene_cover(1, True), 1920, 1080, 8, 50, 100)]
for name, sp, cam, w, h, spp, depth, K in cases:
    with M.State(M.Args(w, h, spp, depth, 1.0), seed=1) as st:
        st.set_world(sp)
        if cam is not None: st.set_camera(cam)
        st.render(3); st.sync()
        c0 = st.read_counters()
        t0 = time.perf_counter(); st.render(K); st.sync(); dt = time.perf_counter() - t0
        c1 = st.read_counters()
        util = (c1["world_hit_calls"] - c0["world_hit_calls"]) / max(1, c1["lane_slots"] - c0["lane_slots"])
        print(f"{name}: {dt / K * 1e3:.3f} ms/frame, {w * h * spp * K / dt * 1e-6:.1f} Msamples/s, lane utilisation {util:.3f}", flush=True)
