#!/usr/bin/env python3
"""Dev helper (GPU box): one mrt_redraw per step on the ctx's own stream vs on a caller's (torch) stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M
mode = sys.argv[1]
stream = None
if mode == "torch":
    import torch
    s = torch.cuda.Stream(); torch.cuda.set_stream(s); stream = s.cuda_stream
for name, sp, cam, w, h, spp, depth in (("C1", M.scene_default(), None, 400, 225, 16, 8), ("interactive", *M.scene_cover(1, True), 1920, 1080, 1, 50)):
    with M.State(M.Args(w, h, spp, depth, 1.0), seed=1, stream=stream) as st:
        st.set_world(sp)
        if cam is not None: st.set_camera(cam)
        for _ in range(20): st.redraw()
        st.sync()
        t0 = time.perf_counter()
        for _ in range(300): st.redraw()
        t1 = time.perf_counter()
        st.sync()
        t2 = time.perf_counter()
        print(f"{mode} {name}: {(t2 - t0) / 300 * 1e3:.3f} ms per redraw (host loop alone {(t1 - t0) / 300 * 1e3:.3f} ms)", flush=True)
