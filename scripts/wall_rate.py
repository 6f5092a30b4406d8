#!/usr/bin/env python3
"""Dev helper (GPU box): pipelined wall-clock Msamples/s of one workload under the tuning switches.
   [MRT_HIER=levels,top] [MRT_BOXES=0|1] [MRT_RNG=1] [MRT_HINT=div,mult] [MRT_READ_EVERY=1] [MRT_SHARD=rank,world] [MRT_STEADY=1]
   python scripts/wall_rate.py scene w h spp steps"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M
a = sys.argv[1:]
scene, w, h, spp, steps = a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4])
sp, cam = (M.scene_cover(1, scene == "cover-glass") if scene.startswith("cover") else M.scene_stress(1, 100) if scene == "stress" else M.scene_stress(1, int(scene[6:])) if scene.startswith("stress") else (M.scene_default(), None))
shard = tuple(int(x) for x in os.environ["MRT_SHARD"].split(",")) if os.environ.get("MRT_SHARD") else None
with M.State(M.Args(w, h, spp, 50, 1.0), seed=1, shard=shard) as st:
    if os.environ.get("MRT_HIER"):
        st.debug_set_hierarchy(*[int(x) for x in os.environ["MRT_HIER"].split(",")])
    if os.environ.get("MRT_BOXES"):
        st.debug_set_boxes(os.environ["MRT_BOXES"] == "1")
    if os.environ.get('MRT_SCHED'):
        a_ = [int(x) for x in os.environ['MRT_SCHED'].split(',')]
        st.debug_set_schedule(a_[0], a_[1])
    if os.environ.get("MRT_SLOTS"):
        st.debug_set_frames_in_flight(int(os.environ["MRT_SLOTS"]))
    if os.environ.get("MRT_HINT"):              # "div,mult": pin the launch schedule
        st.set_schedule_hint(*[int(x) for x in os.environ["MRT_HINT"].split(",")])
    read_every = bool(os.environ.get("MRT_READ_EVERY"))     # a viewer: the framebuffer is read back after every redraw
    st.set_world(sp)
    if cam is not None: st.set_camera(cam)
    if os.environ.get("MRT_RNG"): st.set_rng_mode(int(os.environ["MRT_RNG"]))
    st.set_draw_counting(False)
    for _ in range(int(os.environ.get('MRT_WARMUP', max(2, int(os.environ.get('MRT_SLOTS', '2')))))): st.redraw()      # (the launch-width controller settles within ~3 x the frames in flight)
    st.sync()
    c0 = st.read_counters()
    t0 = time.perf_counter()
    stamps = []
    for _ in range(steps):
        st.redraw()
        stamps.append(time.perf_counter())         # (a call returns when the oldest frame in flight has ended: the back-pressure)
        if read_every: st.read_framebuffer()
    st.sync()
    dt = time.perf_counter() - t0
    c1 = st.read_counters()
    util = (c1["world_hit_calls"] - c0["world_hit_calls"]) / max(1, c1["lane_slots"] - c0["lane_slots"])
    print(f"{scene} {w}x{h}x{spp} HIER={os.environ.get('MRT_HIER')} BOXES={os.environ.get('MRT_BOXES')} RNG={os.environ.get('MRT_RNG')}: "
          f"{w * h * spp * steps / dt * 1e-6 / (shard[1] if shard else 1):.0f} Msamples/s, {dt / steps * 1e3:.1f} ms/step, lane util {util:.3f}, top {c1['sweep_records']}, "
          f"{'read back every frame, ' if read_every else ''}schedule {st.get_schedule()}", flush=True)
    if os.environ.get("MRT_STEADY") and not read_every:
        # the pipelined rate WITHOUT the run's fill and drain: calls are paced by completions, so between the return of call a and
        # the return of the last call exactly (steps - 1 - a) frames have ended, with the pipeline full at both instants
        f = st.get_schedule()["frames_in_flight"]
        a = min(2 * f, steps // 3)
        n_done, span = steps - 1 - a, stamps[-1] - stamps[a]
        print(f"    steady state (frames ending between call {a} and call {steps - 1}, {f} in flight throughout): "
              f"{w * h * spp * n_done / span * 1e-6 / (shard[1] if shard else 1):.0f} Msamples/s per GPU over {span:.1f} s", flush=True)
