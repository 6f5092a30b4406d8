#!/usr/bin/env python3
"""CPU experiment (numpy, no GPU): how many bounds of the stress scene's hierarchy a ray touches per level, under
different culling rules.  Decides what the walk of kernels.hip should prune (DESIGN_HISTORY.md, round 3).

    python experiments/bound_stats.py [n_rays]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import myraytracer_amd as M
from oracle import pyoracle as O
from common import to_oracle_spheres
from test_hierarchy_host import build

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
scene = sys.argv[2] if len(sys.argv) > 2 else "stress"
rng = np.random.default_rng(5)
sc, cam = M.scene_stress(1, 100) if scene == "stress" else M.scene_cover(1, True)
h = build(M, sc)
levels = h["levels"]
print("levels", levels, "top", len(h["top"]), "members", h["n_members"], "direct", h["n_direct"])
packed = O.pack_world(to_oracle_spheres(O, sc))

# camera rays
lf, la = np.array(cam.lookfrom, float), np.array(cam.lookat, float)
fw = (la - lf) / np.linalg.norm(la - lf)
right = np.cross(fw, np.array(cam.vup, float)); right /= np.linalg.norm(right)
up = np.cross(right, fw)
th = np.tan(np.radians(cam.vfov_deg) / 2)
u = rng.uniform(-1, 1, n_rays) * th * 16 / 9
v = rng.uniform(-1, 1, n_rays) * th
d = fw[None] + u[:, None] * right[None] + v[:, None] * up[None]
d /= np.linalg.norm(d, axis=1, keepdims=True)
cam_rays = np.concatenate([np.repeat(lf[None], n_rays, 0), d], 1).astype(np.float32)

centers = np.asarray(sc["center"], float).reshape(-1, 3)
radii = np.asarray(sc["radius"], float)


def bounce(rays):
    hit, t, _, _ = O.world_hit_batch(packed, rays)
    ok = hit >= 0
    r = rays[ok].astype(float)
    p = r[:, :3] + t[ok, None] * r[:, 3:]
    n = (p - centers[hit[ok]]) / radii[hit[ok], None]
    s = rng.normal(size=n.shape); s /= np.linalg.norm(s, axis=1, keepdims=True)
    nd = n + s
    nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    return np.concatenate([p, nd], 1).astype(np.float32)


b1 = bounce(cam_rays)
b2 = bounce(b1)
b3 = bounce(b2)
bounce_rays = np.concatenate([b1, b2, b3])[:n_rays]

# per level: records (centre, R) and AABBs of the member spheres under each node
n_hier = h["direct_first"] if h["n_direct"] else h["n_members"]
mem = h["nodes"][:n_hier].astype(float)
never = np.isinf(mem[:, 3])
mc, mr = mem[:, :3], np.sqrt(np.where(never, 0, -mem[:, 3]))
recs, boxes = {}, {}
bases = h["level_base"]
for k in range(1, levels + 1):
    rr = h["top"] if k == levels else h["nodes"][bases[k]:(bases[k + 1] if k + 1 < levels else len(h["nodes"]))]
    rr = rr.astype(float)
    n_k = (n_hier + 4 ** k - 1) // 4 ** k
    rr = rr[:n_k]
    recs[k] = rr
    lo = np.full((n_k, 3), np.inf); hi = np.full((n_k, 3), -np.inf)
    for j in range(n_k):
        sl = slice(j * 4 ** k, min(n_hier, (j + 1) * 4 ** k))
        ok = ~never[sl]
        if ok.any():
            lo[j] = (mc[sl][ok] - mr[sl][ok, None]).min(0)
            hi[j] = (mc[sl][ok] + mr[sl][ok, None]).max(0)
    boxes[k] = (lo, hi)


def sphere_pass(rays, rec, t_prune=None):
    o, dd = rays[:, None, :3].astype(float), rays[:, None, 3:].astype(float)
    oc = o - rec[None, :, :3]
    b = (oc * dd).sum(-1)
    c = (oc * oc).sum(-1) + rec[None, :, 3]
    disc = b * b - c
    ok = (disc >= 0) & ~((b >= 0) & (c >= 0)) & np.isfinite(rec[None, :, 3])
    if t_prune is not None:       # entry of the bound beyond the known hit
        entry = -b - np.sqrt(np.maximum(disc, 0))
        ok &= ~(entry > t_prune[:, None])
    return ok


def box_pass(rays, box, t_prune=None):
    lo, hi = box
    o, dd = rays[:, None, :3].astype(float), rays[:, None, 3:].astype(float)
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (lo[None] - o) / dd
        t1 = (hi[None] - o) / dd
    tn = np.minimum(t0, t1).max(-1)
    tf = np.maximum(t0, t1).min(-1)
    ok = (tn <= tf) & (tf >= 0) & np.isfinite(lo[None, :, 0])
    if t_prune is not None:
        ok &= ~(tn > t_prune[:, None])
    return ok


QUAD = os.environ.get('QUAD') == '1'
RMIN = 0.1 if scene == 'stress' else 0.2


def sat_pass(rays, box, behind=True):
    """the kernel's line-vs-box test: separating axes d x e_i with the slack terms, optionally + 'not entirely behind'"""
    lo, hi = box
    c = (0.5 * (lo + hi)).astype(np.float32).astype(float)
    e0 = np.maximum(hi - c, c - lo)
    e0 = np.where(np.isfinite(e0), e0, -3e38)
    o, d = rays[:, None, :3].astype(float), rays[:, None, 3:].astype(float)
    p = o - c[None]
    if QUAD:
        k2 = 1.2e-6 / RMIN
        K = k2 * (p * p).sum(-1) + (k2 * (e0 * e0).sum(-1))[None]
    else:
        kpad = 1.4e-3 * np.abs(e0).sum(-1)
        K = 2.0 ** -9 * np.abs(p).sum(-1) + kpad[None]
    ad = np.abs(d)
    ok = np.ones(p.shape[:2], bool)
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        l = p[..., j] * d[..., k] - p[..., k] * d[..., j]
        r = e0[None, :, j] * ad[..., k] + e0[None, :, k] * ad[..., j] + K
        ok &= np.abs(l) <= r
    if behind:
        reach = (e0[None] * ad).sum(-1) + K
        ok &= (reach - (p * d).sum(-1)) >= 0
    return ok & np.isfinite(lo[None, :, 0])


def walk(rays, rule, t_prune=None):
    counts = []
    parent = None
    for k in range(levels, 0, -1):
        if rule == "sphere":
            ok = sphere_pass(rays, recs[k], t_prune)
        elif rule == "box":
            ok = box_pass(rays, boxes[k], t_prune)
        elif rule == "sat":
            ok = sat_pass(rays, boxes[k], True)
        elif rule == "satline":
            ok = sat_pass(rays, boxes[k], False)
        elif rule == "kernel":       # sphere sweep at the top, then SAT (with the behind test) everywhere
            ok = sat_pass(rays, boxes[k], True)
            if k == levels:
                raw = sphere_pass(rays, recs[k], None)
                counts.append(raw.sum(1).mean())
                ok &= raw
        elif rule == "topbox":       # sphere sweep + box filter at the top, spheres below
            ok = sphere_pass(rays, recs[k], None)
            if k == levels:
                counts.append(ok.sum(1).mean())
                ok &= sat_pass(rays, boxes[k], False)
        elif rule == "kernelline":
            ok = sat_pass(rays, boxes[k], False)
            if k == levels:
                raw = sphere_pass(rays, recs[k], None)
                counts.append(raw.sum(1).mean())
                ok &= raw
        else:
            ok = sphere_pass(rays, recs[k], t_prune) & box_pass(rays, boxes[k], t_prune)
        if parent is not None:
            ok &= np.repeat(parent, 4, axis=1)[:, :ok.shape[1]]
        counts.append(ok.sum(1).mean())
        parent = ok
    return counts


for name, rays in (("camera", cam_rays), ("bounce", bounce_rays)):
    hit, t, _, _ = O.world_hit_batch(packed, rays)
    t_final = np.where(hit >= 0, t, 1e4).astype(float)
    # the ground sphere (index 0) alone
    oc = rays[:, :3].astype(float) - centers[0]
    bq = (oc * rays[:, 3:]).sum(1); cq = (oc * oc).sum(1) - radii[0] ** 2
    dq = bq * bq - cq
    tg = np.where(dq >= 0, -bq - np.sqrt(np.maximum(dq, 0)), 1e4)
    tg = np.where(tg > 1e-3, tg, np.where(dq >= 0, -bq + np.sqrt(np.maximum(dq, 0)), 1e4))
    tg = np.where(tg > 1e-3, tg, 1e4)
    print(f"--- {name} rays: {len(rays)}, hit fraction {np.mean(hit >= 0):.2f}, hits on a small sphere {np.mean(hit > 0):.2f}")
    for rule in ("sphere", "box", "both", "sat", "satline", "kernel", "kernelline", "topbox"):
        for pn, tp in ((("no prune", None), ("ground t", tg), ("final t (ideal order)", t_final)) if rule in ("sphere", "box", "both") and os.environ.get("ALL") else (("no prune", None),)):
            c = walk(rays, rule, tp)
            print(f"{rule:7s} {pn:24s} per level top..clusters: " + " ".join(f"{x:6.2f}" for x in c) + f"   total {sum(c):6.2f}")
