#!/usr/bin/env python3
"""Dev helper (GPU box): a long randomised parity campaign, HIP vs oracle, bit for bit.
   python scripts/parity_campaign.py [n_cases] [first_seed] [seconds] [--list]
Random scenes as in tests/test_gpu_random_scenes.py plus far / tiny / clustered / planar layouts, scene sizes up
to a few thousand spheres, random hierarchy depth rules.  Prints one line per failure and a summary."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import myraytracer_amd as M
from oracle import pyoracle as O
from common import gpu_render, oracle_render, mismatch_report

def scene(rng, n):
    sc = np.zeros(n, M.SPHERE_DTYPE)
    layout = rng.integers(0, 5)
    scale = float(10.0 ** rng.uniform(-2, 4)) if rng.random() < 0.4 else 1.0
    off = rng.uniform(-1, 1, 3) * (float(10.0 ** rng.uniform(0, 6)) if rng.random() < 0.3 else 0.0)
    for i in range(n):
        kind = rng.integers(0, 12)
        if layout == 1:      # planar grid, like the cover scene
            c = np.array([rng.uniform(-8, 8), 0.2, rng.uniform(-8, 8)]); r = rng.uniform(0.05, 0.3)
        elif layout == 2:    # tight blobs
            c = rng.integers(-3, 4, 3) * 2.0 + rng.normal(0, 0.15, 3); r = rng.uniform(0.02, 0.4)
        elif layout == 3:    # concentric / coincident
            c = np.array([0.0, 0.0, -3.0]) + (rng.normal(0, 1e-3, 3) if rng.random() < 0.5 else 0); r = 0.3 + 0.002 * (i % 97)
        else:
            c = rng.uniform(-4, 4, 3); r = rng.uniform(0.1, 1.5)
        if kind == 0: c = np.array([rng.uniform(-5, 5), -rng.uniform(50, 2000), rng.uniform(-5, 5)]); r = abs(c[1]) - rng.uniform(0, 1)
        elif kind == 1: r = rng.uniform(1e-3, 2e-2)
        elif kind == 2: r = -r
        ty = int(rng.integers(1, 4))
        param = float(rng.uniform(0, 1.2)) if ty == 2 else float(rng.choice([1.0, 1.33, 1.5, 2.4, 0.7]))
        sc[i] = (tuple(c * scale + off), r * scale, ty, tuple(rng.uniform(0.05, 1.0, 3)), param)
    return sc, scale, off

def case_params(case):
    """Everything case number `case` renders with, drawn from its own generator in a fixed order (the campaign is
    deterministic in the case number; `--list` prints these without touching the GPU)."""
    rng = np.random.default_rng(case)
    n = int(rng.choice([1, 3, 8, 17, 64, 100, 257, 300, 520, 1025, 1100, 1500, 2000, 3000, 5000]))      # (> 1,020: the large-scene layout, with boxes)
    if os.environ.get('MRT_CAMPAIGN_LARGE'):        # only scenes that walk boxes
        n = int(rng.choice([1021, 1100, 1500, 2000, 3000, 4100, 5000, 8000]))
    sc, scale, off = scene(rng, n)
    if rng.random() < 0.25:
        cam = None
    else:
        lf = rng.uniform(-6, 6, 3) * scale + off
        la = rng.uniform(-1, 1, 3) * scale + off
        cam = M.Camera(1, tuple(lf), tuple(la), (0.1 * rng.normal(), 1.0, 0.1 * rng.normal()), float(rng.uniform(5, 100)),
                       float(rng.choice([0.0, 0.0, 0.5, 3.0])), float(rng.uniform(0.5, 12) * scale))
    w, h = int(rng.integers(8, 80)), int(rng.integers(8, 48))
    spp, depth = int(rng.choice([1, 2, 3, 5])), int(rng.choice([1, 2, 5, 13, 50]))
    mode = int(rng.random() < 0.3)                      # counter-RNG mode, sometimes with several blocks of 64 samples
    if mode and rng.random() < 0.4:
        spp, w, h = int(rng.choice([64, 65, 130, 200])), min(w, 24), min(h, 16)
    seed = int(rng.integers(0, 2 ** 62))
    frames = int(rng.choice([1, 1, 2, 5, 9])) if spp <= 5 else 1       # several frames go through mrt_render's batches, or the frame slots
    hier = (int(rng.integers(1, 5)), int(rng.choice([1, 4, 16, 64, 256])))
    lim = max(float(np.abs(sc["center"]).max()), float(np.abs(sc["radius"]).max()))
    p = dict(case=case, n=n, sc=sc, scale=scale, cam=cam, w=w, h=h, spp=spp, depth=depth, mode=mode, seed=seed, frames=frames, hier=hier,
             skip=lim > 5e6)            # the ABI rejects |v| > 1e7
    if p["skip"]:
        return p
    p["sweep"] = int(rng.integers(0, 3))                                 # automatic / VALU / matrix-core sweep
    p["boxes"] = [True, True, True, False, 1][int(rng.integers(0, 5))]   # large scenes: real boxes, or boxes opened wide (they never reject)
    p["batching"] = int(rng.choice([0, 1, 1, 2, 3]))                     # frame by frame / automatic / frames in the lane / frames as queue layers
    p["in_flight"] = int(rng.choice([0, 0, 1, 3, 8]))                    # round 4: automatic / that many frames in flight
    p["count"] = bool(rng.random() < 0.7)
    # round 5 (drawn last, so that the earlier parameters of a case number are what they were): a pinned launch schedule --
    # a frame on 1 / div of the waves, max(2, div) x mult frames in flight -- where the frames in flight are not forced
    p["hint"] = [(0, 0), (0, 0), (1, 1), (2, 2), (4, 2), (8, 2), (8, 1), (4, 1), (1, 8), (2, 4)][int(rng.integers(0, 10))]
    return p

def describe(p):
    if p["skip"]:
        return f"case {p['case']}: skipped (out of the ABI's coordinate range)"
    return (f"case {p['case']}: n={p['n']} {p['w']}x{p['h']}x{p['spp']} depth {p['depth']} rng_mode {p['mode']} frames {p['frames']} hier {p['hier']} "
            f"sweep {p['sweep']} boxes {p['boxes']} batching {p['batching']} frames_in_flight {p['in_flight']} hint {p['hint']} count {int(p['count'])} "
            f"camera {'default' if p['cam'] is None else 'look-at'} scale {p['scale']:.3g}")

def main():
    args = [a for a in sys.argv[1:] if a != "--list"]
    n_cases = int(args[0]) if len(args) > 0 else 200
    first = int(args[1]) if len(args) > 1 else 50000
    budget = float(args[2]) if len(args) > 2 else 1e9          # seconds: stop cleanly after this long
    if "--list" in sys.argv:            # the cases' parameters only: no GPU, no oracle
        for case in range(first, first + n_cases):
            print(describe(case_params(case)))
        return
    fails, skipped, t0 = 0, 0, time.time()
    for case in range(first, first + n_cases):
        p = case_params(case)
        if p["skip"]:
            skipped += 1
            continue
        sc, cam, w, h, spp, depth, mode, seed, frames, count = (p[k] for k in ("sc", "cam", "w", "h", "spp", "depth", "mode", "seed", "frames", "count"))
        cnt = O.Counters()
        ref = oracle_render(O, sc, cam, w, h, spp, depth, seed, frames, 1.0, counters=cnt, rng_mode=mode)
        try:
          with M.State(M.Args(w, h, spp, depth, 1.0), seed=seed) as st:
            st.debug_set_hierarchy(*p["hier"])
            st.debug_set_sweep(p["sweep"])
            st.debug_set_boxes(p["boxes"])
            st.debug_set_frame_batching(p["batching"])
            st.debug_set_frames_in_flight(p["in_flight"])
            if p["hint"] != (0, 0):
                st.set_schedule_hint(*p["hint"])
            st.set_draw_counting(count)
            st.set_world(sc)
            if cam is not None: st.set_camera(cam)
            st.set_rng_mode(mode)
            st.render(frames)
            got, c = st.read_framebuffer(), st.read_counters()
        except M.MrtError as e:          # e.g. MRT_ERR_STALLED: the campaign records it and goes on
            fails += 1
            print(f"FAIL {describe(p)}: {e}", flush=True)
            continue
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        ok = same.all() and c["world_hit_calls"] == cnt.world_hit_calls and c["rng_draws"] == (cnt.rng_draws if count else 0)
        if not ok:
            fails += 1
            print(f"FAIL {describe(p)}: {mismatch_report(got, ref)}", flush=True)
        if (case - first) % 25 == 24:
            print(f"... {case - first + 1} cases, {fails} failures, {time.time() - t0:.0f} s", flush=True)
        if time.time() - t0 > budget:
            n_cases = case - first + 1
            break
    print(f"campaign: {n_cases} cases ({skipped} skipped: out of the ABI's coordinate range), {fails} failures")
    sys.exit(1 if fails else 0)

if __name__ == "__main__":
    main()
