#!/usr/bin/env python3
"""Dev helper (GPU box): does an IDLE context that holds many streams slow another context's frames down?
   python scripts/queue_oversubscription.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M

def rate(tag):
    sp, cam = M.scene_cover(1, False)
    with M.State(M.Args(1200, 675, 64, 50, 1.0), seed=1) as st:
        st.set_world(sp); st.set_camera(cam); st.set_draw_counting(False)
        st.set_schedule_hint(4, 2)
        for _ in range(40): st.redraw()
        st.sync()
        t0 = time.perf_counter()
        for _ in range(200): st.redraw()
        st.sync()
        dt = time.perf_counter() - t0
    print(f"{tag}: C2 at (4, 2): {1200 * 675 * 64 * 200 / dt * 1e-6:.0f} Msamples/s", flush=True)

def idle(hint):
    a = M.State(M.Args(64, 40, 2, 8, 1.0), seed=1)
    a.set_world(M.scene_default())
    a.set_schedule_hint(*hint)
    for _ in range(3): a.redraw()
    a.sync()
    return a

print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
rate("alone")
a = idle((1, 1)); rate("beside an idle context with 2 frame slots"); a.close()
a = idle((2, 2)); rate("beside an idle context that probed (16 slot streams)"); a.close()
rate("after closing it")
a = idle((8, 2)); b = idle((8, 2)); rate("beside two idle contexts with 16 slot streams each"); a.close(); b.close()
rate("after closing them")
