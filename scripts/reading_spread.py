#!/usr/bin/env python3
"""How far may a wgpu backend legitimately sit from this library?  (CPU only; no reference code is run or imported.)

WGSL leaves the lowering of dot() -- and with it the discriminant of sphere_hit, shader.wgsl:277-282 -- to naga's backend:
fused multiply-adds or separately rounded products and sums are both legal.  The oracle (and, bit for bit, the HIP path)
fixes ONE reading, MRT-F32 (fma chains).  This script renders config C1 (the reference's shipped 4-sphere scene,
400 x 225, 16 spp, depth 8, seed 1) with that reading and with the other common one -- no fused operation anywhere
(oracle/librt_oracle_nofma.so, rt_oracle.c ORC_READING_NOFMA) -- from the same seeds, and reports the RMSE between the two
images next to the Monte-Carlo standard error of the image itself (RMSE between two seeds / sqrt 2).  Path tracing is
chaotic in the last bit: a discriminant or range test that flips sends a whole sample down another path, so the spread
between readings is set by how OFTEN that happens, not by 1-ulp differences.

    python scripts/reading_spread.py [--json]        (about 5 s on 8 cores)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def measure(width=400, height=225, spp=16, depth=8, seed=1):
    import myraytracer_amd as M          # host-side scene generators only (no GPU)
    from oracle import pyoracle as O
    from common import oracle_render, rmse_rgb
    scene = M.scene_default()
    a = oracle_render(O, scene, None, width, height, spp, depth, seed)
    with O.reading("nofma"):
        b = oracle_render(O, scene, None, width, height, spp, depth, seed)
    a2 = oracle_render(O, scene, None, width, height, spp, depth, seed + 1)
    differ = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
    d = np.abs(a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64))
    # pixels where a whole sample took another path (a difference no rounding of one sample's colour explains)
    jumped = d.max(axis=-1) > 1e-3 / spp
    return {
        "config": f"C1: shipped 4-sphere scene, {width}x{height}, {spp} spp, depth {depth}, seed {seed}",
        "readings": ["MRT-F32 (fma chains; the oracle and the HIP path)", "no fused operation anywhere (dot = x*x + y*y + z*z)"],
        "rmse_between_readings": rmse_rgb(a, b),
        "monte_carlo_standard_error": rmse_rgb(a, a2) / np.sqrt(2.0),
        "pixels_differing_in_any_bit": float(differ.mean()),
        "pixels_with_a_diverged_sample": float(jumped.mean()),
        "max_abs_difference": float(d.max()),
        "median_abs_difference_of_differing_pixels": float(np.median(d.max(axis=-1)[differ])) if differ.any() else 0.0,
    }


if __name__ == "__main__":
    r = measure()
    if "--json" in sys.argv:
        print(json.dumps(r, indent=1))
    else:
        for k, v in r.items():
            print(f"{k:45s} {v:.4e}" if isinstance(v, float) else f"{k:45s} {v}")
        print(f"ratio: the two readings differ by {r['rmse_between_readings'] / r['monte_carlo_standard_error']:.3f} x the image's own "
              f"Monte-Carlo standard error; north_star's RMSE < 1e-4 is "
              f"{'met' if r['rmse_between_readings'] < 1e-4 else 'NOT met'} between two LEGAL readings of the reference")
