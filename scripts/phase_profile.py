#!/usr/bin/env python3
"""Dev helper (GPU box): phase shares of the bounce loop from the -DMRT_STAMPS build.
   MRT_LIB_OVERRIDE=myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/phase_profile.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M
from myraytracer_amd import _lib
a = sys.argv[1:]
scene = a[0] if a else "cover-glass"
w, h, spp = (int(a[1]), int(a[2]), int(a[3])) if len(a) > 3 else (1920, 1080, 32)
frames = int(a[4]) if len(a) > 4 else 1      # > 1: one mrt_render(frames), i.e. frame batches where they apply
sp, cam = (M.scene_cover(1, scene == "cover-glass") if scene.startswith("cover") else M.scene_stress(1, 100) if scene == "stress"
           else (M.scene_default(), None))
with M.State(M.Args(w, h, spp, 50, 1.0), seed=1) as st:
    st.set_world(sp)
    if cam is not None: st.set_camera(cam)
    st.render(frames); st.sync()
    raw = (C.c_uint64 * 16)()
    _lib.load().mrt_debug_read_counters(st._ctx, raw)
    names = ["tail: camera rays, hit records, rejection loop, scatter, normalize", "sweep", "walk: node rounds", "walk: root rounds",
             "paths ending without a scatter", "release/refill/acquire"]
    ph = [raw[6 + k] for k in range(6)] + [raw[5]]
    names = names + ["walk: owners unpack masks into items"]
    tot = sum(ph)
    print("kernel ms", st.last_kernel_ms(), "wave sweeps", raw[3] / 64)
    for n, v in zip(names, ph):
        print(f"{n:70s} {v / max(1, tot):7.3%}  cycles/wave-iteration {v / (raw[3] / 64):9.1f}")
    sweeps = raw[3] / 64
    print(f"node rounds/sweep {raw[12] / sweeps:.2f} (items/round {raw[14] / max(1, raw[12]):.1f}), root rounds/sweep {raw[13] / sweeps:.2f} (items/round {raw[15] / max(1, raw[13]):.1f})")
