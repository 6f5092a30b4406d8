#!/usr/bin/env python3
"""Dev helper (GPU box): render one frame and print kernel time + counters."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M

def run(scene, w, h, spp, depth=50, frames=int(os.environ.get('MRT_FRAMES', '2')), shard=None):
    if scene == "cover-glass": sp, cam = M.scene_cover(1, True)
    elif scene == "cover": sp, cam = M.scene_cover(1, False)
    elif scene == "stress": sp, cam = M.scene_stress(1, 100)
    else: sp, cam = M.scene_default(), None
    with M.State(M.Args(w, h, spp, depth, 1.0), seed=1, shard=shard) as st:
        sched = os.environ.get("MRT_SCHED")        # "pilot,waves_per_cu"
        if sched:
            from myraytracer_amd import _lib
            a_ = [int(x) for x in sched.split(",")]
            assert _lib.load().mrt_debug_set_schedule(st._ctx, a_[0], a_[1]) == 0
        if os.environ.get("MRT_SWEEP"):       # 1 = SGPR-fed VALU sweep, 2 = matrix-core sweep
            from myraytracer_amd import _lib
            assert _lib.load().mrt_debug_set_sweep(st._ctx, int(os.environ["MRT_SWEEP"])) == 0
        if os.environ.get("MRT_HIER"):        # "max_levels,top_target"
            from myraytracer_amd import _lib
            h_ = [int(x) for x in os.environ["MRT_HIER"].split(",")]
            assert _lib.load().mrt_debug_set_hierarchy(st._ctx, h_[0], h_[1]) == 0
        if os.environ.get("MRT_BOXES"):
            st.debug_set_boxes(os.environ["MRT_BOXES"] == "1")
        if os.environ.get("MRT_CLUSTER"):
            from myraytracer_amd import _lib
            _lib.load().mrt_debug_set_cluster_factor(st._ctx, float(os.environ["MRT_CLUSTER"]))
        st.set_world(sp)
        if cam is not None: st.set_camera(cam)
        st.render(1); st.sync()
        c0 = st.read_counters()
        st.render(frames - 1); st.sync()
        c1 = st.read_counters()
        ms = st.kernel_ms_history(frames)[1:]
        d = {k: (c1[k] - c0[k]) / (frames - 1) for k in c0 if k != "sweep_records"}
        n = len(sp)
        ms_avg = sum(ms) / len(ms)
        tests = d["world_hit_calls"] * n
        share = 1.0 / shard[1] if shard else 1.0
        print(json.dumps({"scene": scene, "n": n, "w": w, "h": h, "spp": spp, "shard": shard, "sched": os.environ.get("MRT_SCHED"), "ms": round(ms_avg, 3), "ms_min": round(min(ms), 3),
                          "Msamples/s": round(w * h * spp * share / ms_avg * 1e-3, 1),
                          "bounces/sample": round(d["world_hit_calls"] / d["samples"], 3),
                          "lane_util": round(d["world_hit_calls"] / max(1, d["lane_slots"]), 4),
                          "Gtests/s": round(tests / ms_avg * 1e-6, 1),
                          "wave_sweeps": d["lane_slots"] / 64, "sweep_records": c1["sweep_records"],
                          "member_tests/bounce": round(d["member_tests"] / d["world_hit_calls"], 2)}))

if __name__ == "__main__":
    a = sys.argv[1:]
    run(a[0] if a else "cover-glass", int(a[1]) if len(a) > 1 else 1920, int(a[2]) if len(a) > 2 else 1080,
        int(a[3]) if len(a) > 3 else 32, int(a[4]) if len(a) > 4 else 50,
        shard=(int(a[5]), int(a[6])) if len(a) > 6 else None)
