#!/usr/bin/env python3
"""Dev helper (GPU box): 1,000,001 spheres (the ABI's limit is 2^20) against the oracle at 32x18: hierarchy build,
16 sweep blocks at the top level, four levels below."""
import os
import sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import myraytracer_amd as M
from oracle import pyoracle as O
from common import gpu_render, oracle_render, mismatch_report
sc, cam = M.scene_stress(5, 1000)
print(len(sc))
t0=time.time(); ref = oracle_render(O, sc, cam, 32, 18, 1, 4, 3); print("oracle", round(time.time()-t0,1))
t0=time.time(); got, c, ms = gpu_render(M, sc, cam, 32, 18, 1, 4, 3); print("gpu", round(time.time()-t0,1), ms, c["sweep_records"])
print(mismatch_report(got, ref))
