#!/usr/bin/env python3
"""Dev helper (GPU box, -DMRT_STAMPS build via MRT_LIB_OVERRIDE): where the wave slots of a pipelined workload go.
   MRT_LIB_OVERRIDE=myraytracer_amd/lib/libmyraytracer_amd_stamps.so python scripts/shard_occupancy.py <scene> <w> <h> <spp> <rank> <world> [frames]
From the per-wave log {start, end} of every launch (a ring over the last 32 frames): over a steady-state window,
   running  = wave slots whose wave is between its first and last loop trip
   dead     = slots of a RESIDENT workgroup whose wave has ended (a workgroup's LDS and wave slots are held until its LAST wave ends)
   empty    = slots no workgroup holds (launch gaps, the next launch not resident yet)
and per launch its duration, its dead share and the gap to the next launch of the same frame slot."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import myraytracer_amd as M
from myraytracer_amd import _lib

scene = sys.argv[1]
w, h, spp, rank, world = (int(x) for x in sys.argv[2:7])
K = min(int(sys.argv[7]) if len(sys.argv) > 7 else 24, 32)
GROUP = int(os.environ.get("MRT_GROUP_WAVES", "4"))
sp, cam = M.scene_stress(1, 100) if scene == "stress" else M.scene_cover(1, True)
L = _lib.load()
with M.State(M.Args(w, h, spp, 50, 1.0), seed=1, shard=(rank, world) if world > 1 else None) as st:
    st.set_world(sp); st.set_camera(cam)
    st.set_draw_counting(False)
    if os.environ.get("MRT_HINT"):
        st.set_schedule_hint(*[int(x) for x in os.environ["MRT_HINT"].split(",")])
    n = C.c_size_t()
    assert L.mrt_debug_wave_log_frame(st._ctx, 0, None, 0, C.byref(n)) == 0
    for _ in range(int(os.environ.get("MRT_WARMUP", "24"))): st.redraw()
    st.sync()
    for _ in range(K): st.redraw()
    st.sync()
    sch = st.get_schedule()
    logs = []
    for back in range(K - 1, -1, -1):                   # oldest first
        log = np.zeros((n.value, 4), np.uint64)
        assert L.mrt_debug_wave_log_frame(st._ctx, back, log.ctypes.data, n.value, C.byref(n)) == 0
        logs.append(log)
lay = (C.c_uint32 * 3)()
frames = []
for log in logs:
    t0, t1 = log[:, 0].astype(np.int64), log[:, 1].astype(np.int64)
    used = t0 != 0
    if not used.any():
        continue
    idx = np.nonzero(used)[0]
    g = idx // GROUP
    gs = np.full(g.max() + 1, np.iinfo(np.int64).max); ge = np.zeros(g.max() + 1, np.int64); gn = np.zeros(g.max() + 1, np.int64)
    np.minimum.at(gs, g, t0[idx]); np.maximum.at(ge, g, t1[idx]); np.add.at(gn, g, 1)
    frames.append(dict(t0=t0[idx], t1=t1[idx], g=g, gs=gs, ge=ge, gn=gn, trips=log[idx, 2].astype(np.int64), hits=log[idx, 3].astype(np.int64)))
if not frames:
    sys.exit("no wave log: is MRT_LIB_OVERRIDE the -DMRT_STAMPS build?")
tick_ms = 1e-5                                          # 100 MHz
slots_in_flight = sch["frames_in_flight"]
T0 = max(f["t0"].min() for f in frames[:min(slots_in_flight, len(frames))])           # every slot has a frame running
T1 = min(f["t1"].max() for f in frames[-min(slots_in_flight, len(frames)):])          # ... and still has
if T1 <= T0: T0, T1 = min(f["t0"].min() for f in frames), max(f["t1"].max() for f in frames)
def overlap(a, b): return np.clip(np.minimum(b, T1) - np.maximum(a, T0), 0, None).sum()
running = sum(overlap(f["t0"], f["t1"]) for f in frames)
held = sum((np.clip(np.minimum(f["ge"], T1) - np.maximum(f["gs"], T0), 0, None) * GROUP).sum() for f in frames)
# the wave slots the chip holds for this kernel: MRT_WAVE_SLOTS (C5's large-scene kernel: 256 CUs x 16), else the steady launches' width x div
wave_slots = int(os.environ.get("MRT_WAVE_SLOTS", "0")) or int(np.median([len(f["t0"]) for f in frames[len(frames) // 2:]])) * sch["div"]
cap = wave_slots * (T1 - T0)
busy_lanes = sum(f["hits"].sum() for f in frames); lane_slots = sum(64 * f["trips"].sum() for f in frames)
print(f"{scene} {w}x{h}x{spp} shard {rank}/{world}: schedule div {sch['div']} x {sch['mult']}, {slots_in_flight} frames in flight, "
      f"{len(frames)} launches logged, workgroups of {GROUP} waves, {wave_slots} wave slots")
print(f"steady window {(T1 - T0) * tick_ms:.1f} ms: wave slots running {running / cap:.3f}, dead (wave ended, its workgroup resident) "
      f"{(held - running) / cap:.3f}, empty {1 - held / cap:.3f}; lane utilisation of the running waves {busy_lanes / max(1, lane_slots):.3f}")
print(" launch  waves  start ms    dur ms  dead share  wave-duration p10/p50/p90 ms      gap to the slot's next launch ms")
base = min(f["t0"].min() for f in frames)
for i, f in enumerate(frames):
    dur = f["t1"] - f["t0"]
    span = f["t1"].max() - f["t0"].min()
    dead = (f["ge"][f["g"]] - f["t1"]).sum() + ((GROUP - f["gn"]) * (f["ge"] - f["gs"]))[f["gn"] > 0].sum()
    heldf = ((f["ge"] - f["gs"]) * GROUP)[f["gn"] > 0].sum()
    nxt = frames[i + slots_in_flight]["t0"].min() - f["t1"].max() if i + slots_in_flight < len(frames) else float("nan")
    print(f"  {i:3d}  {len(dur):6d}  {(f['t0'].min() - base) * tick_ms:8.2f}  {span * tick_ms:8.2f}  {dead / max(1, heldf):10.3f}  "
          f"{np.percentile(dur, 10) * tick_ms:8.2f} {np.percentile(dur, 50) * tick_ms:8.2f} {np.percentile(dur, 90) * tick_ms:8.2f}      {nxt * tick_ms:8.2f}")
