#!/usr/bin/env python3
"""Row fixtures at BASELINE.json's REAL sample counts (C3 512 spp, C4 1,024 spp, C5 4,096 spp), generated with
the CPU oracle: tests/golden/fullsize_rows.npz + fullsize_rows.json.

One lane of the HIP kernel runs a pixel's whole sequential Xoshiro128+ chain (shader.wgsl:377-382), so the
chain length -- spp x path length -- is exactly the dimension a reduced-spp test does not exercise.  The oracle
needs minutes per C5 row (10,001 spheres x 4,096 spp), too long for a test on the GPU box, so its rows are
computed once here and committed; tests/test_gpu_fullspp.py compares the GPU frame's same rows bit for bit,
and the rows' world_hit_calls with the GPU's per-pixel costs.  C3 additionally gets the whole frame's
counters (samples, world_hit_calls, rng_draws), which is what bench.py's headline run accumulates per step.

Like every vector under tests/golden/ these pin the ORACLE; parity with the reference's own floating point
is unpinned (oracle/rt_oracle.h).

Run from the repo root:  python tests/golden/make_fullsize_rows.py [--threads N] [--only c3,c4,c5,c5ctr]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import myraytracer_amd as M          # host-only scene builders
from oracle import pyoracle as O
from common import to_oracle_camera, to_oracle_spheres

GOLDEN = os.path.dirname(os.path.abspath(__file__))

# name -> scene, width, height, spp, depth, seed, frames, rows, whole-frame counters?
CONFIGS = {
    "c3": dict(scene="cover-glass", width=1920, height=1080, spp=512, depth=50, seed=1, frames=2,
               rows=[3, 540, 1000], whole_frame_counters=True),
    "c4": dict(scene="cover-glass", width=3840, height=2160, spp=1024, depth=50, seed=1, frames=1,
               rows=[5, 1080, 2000], whole_frame_counters=False),
    "c5": dict(scene="stress", width=1920, height=1080, spp=4096, depth=50, seed=1, frames=1,
               rows=[10, 500, 900], whole_frame_counters=False),
    # the counter-RNG extension (blocks of 64 samples, summed blockwise): what `bench.py --config c5 --rng counter`
    # and the 1/8 shares of an 8-GPU C5 run execute (64 block layers per pixel)
    "c5ctr": dict(scene="stress", width=1920, height=1080, spp=4096, depth=50, seed=1, frames=1,
                  rows=[10, 500, 900], whole_frame_counters=False, rng_mode=1),
}


def scene_of(name):
    if name == "cover-glass":
        return M.scene_cover(1, True)
    if name == "stress":
        return M.scene_stress(1, 100)
    raise ValueError(name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--only", default="c3,c4,c5,c5ctr")
    a = ap.parse_args()
    npz_path, json_path = os.path.join(GOLDEN, "fullsize_rows.npz"), os.path.join(GOLDEN, "fullsize_rows.json")
    arrays = dict(np.load(npz_path)) if os.path.exists(npz_path) else {}
    meta = json.load(open(json_path)) if os.path.exists(json_path) else {
        "generator": "tests/golden/make_fullsize_rows.py", "oracle": "oracle/rt_oracle.c (MRT-F32)",
        "parity_with_reference": "unpinned", "configs": {}}
    for name in a.only.split(","):
        cfg = CONFIGS[name]
        spheres, cam = scene_of(cfg["scene"])
        packed = O.pack_world(to_oracle_spheres(O, spheres))
        ocam = to_oracle_camera(O, cam)
        w, h, spp, depth, seed = cfg["width"], cfg["height"], cfg["spp"], cfg["depth"], cfg["seed"]
        seeds = O.fill_seeds(seed, w, h)
        entry = dict(cfg)
        entry["n_spheres"] = int(len(spheres))
        entry["row_counters"] = []
        fb = np.zeros((h, w, 4), np.float32)
        for f in range(cfg["frames"]):
            nxt = np.zeros_like(fb)
            per_row = []
            for y in cfg["rows"]:
                t0 = time.time()
                c = O.Counters()
                out = O.render_frame(w, h, spp, depth, packed, ocam, seeds, O.frame_shuffle(seed, f),
                                     O.frame_weight(f, 1.0), fb, rows=(y, y + 1), nthreads=a.threads, counters=c,
                                     rng_mode=cfg.get("rng_mode", 0))
                nxt[y] = out[y]
                arrays[f"{name}_f{f}_row{y}"] = out[y].copy()
                d = c.as_dict()
                per_row.append({"row": y, "samples": d["samples"], "world_hit_calls": d["world_hit_calls"],
                                "rng_draws": d["rng_draws"]})
                print(f"{name} frame {f} row {y}: {time.time() - t0:.1f} s, {d['world_hit_calls'] / d['samples']:.3f} bounces/sample",
                      flush=True)
            entry["row_counters"].append(per_row)
            fb = nxt
        if cfg["whole_frame_counters"]:
            t0 = time.time()
            c = O.Counters()
            O.render_frame(w, h, spp, depth, packed, ocam, seeds, O.frame_shuffle(seed, 0), 0.0, None,
                           nthreads=a.threads, counters=c, rng_mode=cfg.get("rng_mode", 0))
            d = c.as_dict()
            entry["frame0_counters"] = {k: d[k] for k in ("samples", "world_hit_calls", "rng_draws")}
            print(f"{name} whole frame 0: {time.time() - t0:.1f} s, {entry['frame0_counters']}", flush=True)
        meta["configs"][name] = entry
        np.savez_compressed(npz_path, **arrays)
        json.dump(meta, open(json_path, "w"), indent=1)


if __name__ == "__main__":
    main()
