#!/usr/bin/env python3
"""Generates the committed fixtures in tests/golden/ with the CPU oracle.

The reference (zetanumbers/myraytracer) has no tests, golden images or fixed seeds, and
cannot be run here, so these vectors pin the ORACLE (regression) and give the GPU tests
inputs + expected outputs that do not depend on the scene generators.  Parity with the
reference's own floating-point results is unpinned (see oracle/rt_oracle.h).

Run from the repo root:  python tests/golden/make_golden.py
Scene inputs come from the product's host-only scene builders (mrt_scene_default /
mrt_scene_cover, no GPU needed); they are stored next to the outputs.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import myraytracer_amd as M          # host-only scene builders
from oracle import pyoracle as O
from make_golden_cases import CASES, GOLDEN, render_case


def main():
    out_cases = []
    for case in CASES:
        case = dict(case)
        if case["scene"] == "default":
            spheres, cam = M.scene_default(), M.Camera()
        else:
            spheres, cam = M.scene_cover(1, case["scene"] == "cover-glass")
        case["spheres_file"] = case["name"] + ".spheres.bin"
        spheres.tofile(os.path.join(GOLDEN, case["spheres_file"]))
        case["camera"] = dict(mode=cam.mode, lookfrom=list(map(float, cam.lookfrom)), lookat=list(map(float, cam.lookat)),
                              vup=list(map(float, cam.vup)), vfov_deg=float(cam.vfov_deg),
                              defocus_angle_deg=float(cam.defocus_angle_deg), focus_dist=float(cam.focus_dist))
        fb, counters = render_case(O, case)
        case["file"] = case["name"] + ".rgba32f.bin"
        fb.tofile(os.path.join(GOLDEN, case["file"]))
        case["counters"] = counters
        case["mean_rgb"] = [float(x) for x in fb[..., :3].mean(axis=(0, 1))]
        out_cases.append(case)
        print(case["name"], fb.shape, case["mean_rgb"], counters["world_hit_calls"])
    json.dump({"generator": "tests/golden/make_golden.py", "oracle": "oracle/rt_oracle.c (MRT-F32)",
               "parity_with_reference": "unpinned", "cases": out_cases},
              open(os.path.join(GOLDEN, "golden.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
