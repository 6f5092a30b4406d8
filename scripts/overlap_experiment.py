#!/usr/bin/env python3
"""Dev experiment (GPU box): do two independent render streams overlap and hide each other's tail?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M
sp, cam = M.scene_cover(1, True)
def mk():
    st = M.State(M.Args(1920, 1080, 512, 50, 1.0), seed=1)
    st.set_world(sp); st.set_camera(cam); st.render(1); st.sync()
    return st
a, b = mk(), mk()
K = 3
t0 = time.perf_counter(); a.render(K); a.sync(); t1 = time.perf_counter()
print("single ctx: %.1f ms/frame" % ((t1 - t0) / K * 1e3))
t0 = time.perf_counter()
for _ in range(K):
    a.redraw(); b.redraw()
a.sync(); b.sync()
t1 = time.perf_counter()
print("two ctxs interleaved: %.1f ms/frame (aggregate)" % ((t1 - t0) / (2 * K) * 1e3))
