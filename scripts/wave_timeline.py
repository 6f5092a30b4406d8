#!/usr/bin/env python3
"""Dev helper (GPU box, -DMRT_STAMPS build): how many waves are resident over the kernel's lifetime."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M
from myraytracer_amd import _lib
a = sys.argv[1:]
w, h, spp = (int(a[0]), int(a[1]), int(a[2])) if len(a) > 2 else (1920, 1080, 32)
sp, cam = M.scene_cover(1, True)
L = _lib.load()
with M.State(M.Args(w, h, spp, 50, 1.0), seed=1) as st:
    st.set_world(sp); st.set_camera(cam)
    n = C.c_size_t()
    L.mrt_debug_wave_log(st._ctx, None, 0, C.byref(n))
    st.render(1); st.sync()
    log = np.zeros((n.value, 4), np.uint64)
    L.mrt_debug_wave_log(st._ctx, log.ctypes.data, n.value, C.byref(n))
    t0, t1, trips = log[:, 0].astype(np.int64), log[:, 1].astype(np.int64), log[:, 2]
    start = t0.min(); t0 -= start; t1 -= start
    total = t1.max()
    print("kernel ms", st.last_kernel_ms(), "span ticks", total, "waves", len(log))
    edges = np.linspace(0, total, 21)
    for i in range(20):
        mid = 0.5 * (edges[i] + edges[i + 1])
        resident = int(((t0 <= mid) & (t1 > mid)).sum())
        print(f"{i*5:3d}%  resident waves {resident:6d}")
    dur = (t1 - t0)
    print("wave duration ticks: mean %.0f  p50 %.0f  p99 %.0f  max %.0f" % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 99), dur.max()))
    print("starts: p50 %.3f p90 %.3f max %.3f of span; ends: p10 %.3f p50 %.3f p90 %.3f" % (
        np.percentile(t0, 50) / total, np.percentile(t0, 90) / total, t0.max() / total,
        np.percentile(t1, 10) / total, np.percentile(t1, 50) / total, np.percentile(t1, 90) / total))
    print("trips per wave: min %d p50 %d max %d" % (trips.min(), np.percentile(trips, 50), trips.max()))
    _, _, rows, width = st.shard_info()
    pc = np.zeros(rows * width, np.uint32)
    L.mrt_debug_read_pixel_costs(st._ctx, pc.ctypes.data, pc.size)
    pc = pc.reshape(rows, width)
    print("pixel cost (trips): mean %.1f p50 %d p99 %d p99.9 %d max %d ; total/6144/64 = %.0f trips per wave slot" % (
        pc.mean(), np.percentile(pc, 50), np.percentile(pc, 99), np.percentile(pc, 99.9), pc.max(), pc.sum() / 6144 / 64))
    ys, xs = np.unravel_index(np.argsort(pc, axis=None)[-5:], pc.shape)
    print("heaviest pixels (x,y,cost):", [(int(x), int(y), int(pc[y, x])) for x, y in zip(xs, ys)])
    last = np.argsort(t1)[-3:]
    for i in last:
        print("last waves: trips=%d start=%.3f end=%.3f" % (trips[i], t0[i] / total, t1[i] / total))
