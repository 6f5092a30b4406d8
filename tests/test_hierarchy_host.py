"""Host-side invariants of the bounding-sphere hierarchy and of the matrix-core operand it is re-expressed in
(api.cpp build_clusters / build_hierarchy / build_top_mfma), checked without a GPU through
mrt_debug_build_hierarchy.  These are the facts the conservativeness argument of DESIGN.md §4 rests on."""
import ctypes as C

import numpy as np
import pytest

from myraytracer_amd import _lib

INFLATE = 1.015          # mrt_internal.h kBoundInflate
SLACK = 2.0 ** -13       # api.cpp kMfmaSlack


def build(mrt, sc, max_levels=4, top_target=256):
    L = _lib.load()
    sc = np.ascontiguousarray(sc, mrt.SPHERE_DTYPE)
    info = (C.c_uint32 * 10)()
    assert L.mrt_debug_build_hierarchy(sc.ctypes.data, len(sc), max_levels, top_target, None, 0, None, 0, None, 0, None, 0, None, info) == 0
    levels, n_top, n_nodes, n_members = info[0], info[1], info[2], info[3]
    top = np.zeros((n_top, 4), np.float32)
    nodes = np.zeros((n_nodes, 4), np.float32)
    midx = np.zeros(n_members, np.uint32)
    mf = np.zeros(n_top // 32 * 512, np.uint16)
    org = (C.c_float * 3)()
    assert L.mrt_debug_build_hierarchy(sc.ctypes.data, len(sc), max_levels, top_target, top.ctypes.data, len(top), nodes.ctypes.data,
                                       len(nodes), midx.ctypes.data, len(midx), mf.ctypes.data, len(mf), org, info) == 0
    return dict(levels=levels, top=top, nodes=nodes, midx=midx, mfma=mf, origin=np.array(list(org), np.float64),
                n_direct=info[4], direct_first=info[5], level_base=[info[6 + k] for k in range(4)], n_members=n_members)


def scenes(mrt):
    rng = np.random.default_rng(77)
    big = np.zeros(2300, mrt.SPHERE_DTYPE)
    for i in range(len(big)):
        big[i] = (tuple(rng.uniform(-30, 30, 3)), float(rng.uniform(0.05, 0.6)) * (-1 if i % 50 == 0 else 1), 1, (0.5, 0.5, 0.5), 0.0)
    big[7] = ((0, -900, 0), 880.0, 1, (0.5, 0.5, 0.5), 0.0)
    yield "default", mrt.scene_default()
    yield "cover", mrt.scene_cover(1, True)[0]
    yield "stress 40x40", mrt.scene_stress(3, 40)[0]
    yield "random 2300", big
    yield "single", mrt.scene_default()[:1]
    yield "empty", np.zeros(0, mrt.SPHERE_DTYPE)


def test_every_sphere_is_a_member_exactly_once_and_levels_are_well_formed(mrt):
    for name, sc in scenes(mrt):
        for max_levels, target in [(4, 256), (4, 1), (1, 64), (2, 8)]:
            h = build(mrt, sc, max_levels, target)
            never = np.isinf(h["nodes"][:h["n_members"], 3])
            real = h["midx"][~never]
            assert sorted(real.tolist()) == list(range(len(sc))), (name, max_levels, target)
            # member records are bit copies of the spheres
            c = np.asarray(sc["center"], np.float32).reshape(-1, 3)
            r = np.asarray(sc["radius"], np.float32)
            mem = h["nodes"][:h["n_members"]][~never]
            assert np.array_equal(mem[:, :3], c[real]) and np.array_equal(mem[:, 3], -(r[real] * r[real])), name
            assert len(h["top"]) % 32 == 0 and h["n_members"] % 4 == 0 and 1 <= h["levels"] <= max_levels
            if len(sc) + 4 <= 1024:
                assert h["levels"] == 1                                       # small scenes keep one level
            bases = h["level_base"][:h["levels"]] + [len(h["nodes"])]
            for k in range(h["levels"]):
                assert (bases[k + 1] - bases[k]) % 4 == 0, (name, k)


def test_every_bound_encloses_the_spheres_under_it_with_its_margin(mrt):
    """node j of level k covers the members [j 4^k, (j+1) 4^k) of the hierarchy part of level 0; its radius must be
    >= 1.015 x the farthest member surface measured from its (f32) centre -- at every level, and for the top."""
    for name, sc in scenes(mrt):
        if len(sc) == 0:
            continue
        for max_levels, target in [(4, 256), (4, 1), (2, 8)]:
            h = build(mrt, sc, max_levels, target)
            c = np.asarray(sc["center"], np.float64).reshape(-1, 3)
            r = np.abs(np.asarray(sc["radius"], np.float64))
            n_hier = h["direct_first"] if h["n_direct"] else h["n_members"]
            never0 = np.isinf(h["nodes"][:n_hier, 3])
            bases = h["level_base"][:h["levels"]] + [None]
            for k in range(1, h["levels"] + 1):
                recs = h["top"] if k == h["levels"] else h["nodes"][bases[k]:bases[k + 1] if k + 1 < h["levels"] else len(h["nodes"])]
                span = 4 ** k
                for j, rec in enumerate(recs):
                    lo, hi = j * span, min(n_hier, (j + 1) * span)
                    ids = h["midx"][lo:hi][~never0[lo:hi]] if lo < hi else np.zeros(0, np.uint32)
                    if len(ids) == 0:
                        assert np.isinf(rec[3]) and rec[3] > 0, (name, k, j)      # never-hit padding
                        continue
                    R = np.sqrt(-np.float64(rec[3]))
                    far = (np.linalg.norm(c[ids] - rec[:3].astype(np.float64), axis=1) + r[ids]).max()
                    assert R >= INFLATE * far * (1 - 1e-6), (name, max_levels, target, k, j, R, far)
            # the direct spheres are the largest ones and sit outside every bound's bookkeeping
            if h["n_direct"]:
                d_ids = h["midx"][h["direct_first"]:h["direct_first"] + h["n_direct"]]
                assert r[d_ids].min() > 8 * np.median(r)


def bf16(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def test_matrix_core_operand_restates_the_top_level_with_its_slack(mrt):
    """Per tile of 32 records the A operand holds C - origin split into bf16 hi + lo (twice the hi part), (1,1,1)
    and Ck in three pieces, rows in MFMA result-register order; Ck must not exceed C.C - R^2 - 2^-13 (C.C + R^2)."""
    for name, sc in scenes(mrt):
        h = build(mrt, sc)
        top, mf, org = h["top"], h["mfma"].reshape(-1, 2, 32, 8), h["origin"]
        for t in range(len(top) // 32):
            for m in range(32):
                rec = top[32 * t + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3)]
                k = np.concatenate([mf[t, 0, m], mf[t, 1, m]])              # the row's 16 K slots
                hi, hi2, lo, ones, ck3, pad = bf16(k[0:3]), bf16(k[3:6]), bf16(k[6:9]), bf16(k[9:12]), bf16(k[12:15]), k[15]
                assert np.array_equal(hi, hi2) and np.array_equal(ones, [1.0, 1.0, 1.0]) and pad == 0
                crel = (rec[:3].astype(np.float64) - org).astype(np.float32).astype(np.float64)
                assert np.all(np.abs(hi + lo - crel) <= 2.0 ** -16 * np.abs(crel) + 1e-300), (name, t, m)
                if np.isinf(rec[3]):
                    assert ck3.sum() > 1e38                                   # never-hit: finite and huge
                    continue
                c2 = float(crel @ crel)
                R2 = -float(rec[3])
                assert ck3.sum() <= c2 - R2 - SLACK * (c2 + R2) + 1e-9 * (c2 + R2), (name, t, m)
                assert ck3.sum() >= c2 - R2 - 1.01 * SLACK * (c2 + R2) - 1e-5 * (np.sqrt(c2 * R2) + R2), (name, t, m)


def bf16_round(x):
    """round-to-nearest-even of float32 values to bf16, returned as float32 (kernels.hip bf16_round)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    a = bf16_round(x)
    b = bf16_round(x - a)
    return a, b, bf16_round((x - a) - b)


def test_matrix_core_sweep_algebra_is_conservative_and_drops_only_what_lies_behind(mrt):
    """The sweep's two GEMMs (kernels.hip mfma_ray_operands / mfma_sweep_tile) restated in numpy from the very A operand the
    host uploads and the scale factors it passes (mrt_debug_mfma_scale): g = A.B1 = -K oc.ds, C = clamp(g |g|) = max(g, 0)^2,
    result = A.B2 + C = K^2 (max(-oc.ds, 0)^2 - U - o.o), candidate = result not < 0.  Products of bf16 pieces are exact in
    float64 here (the hardware accumulates in f32; that error is what the 2^-13 slack is for).  Every top-level bound whose
    ENCLOSED sphere (radius / 1.015) the ray's forward half-line touches must be a candidate; every bound that is dropped
    although the full line touches it must lie entirely behind the origin; and |g| must stay below 1/2."""
    L = _lib.load()
    rng = np.random.default_rng(9)
    for name, sc in scenes(mrt):
        if len(sc) < 30:
            continue
        h = build(mrt, sc)
        top, org = h["top"].astype(np.float64), h["origin"]
        A = bf16(h["mfma"].reshape(-1, 2, 32, 8))                              # [tile, half, row, 8] -> values
        A = np.concatenate([A[:, 0], A[:, 1]], axis=-1)                        # [tile, row, 16]
        order = [32 * t + 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3) for t in range(len(top) // 32) for m in range(32)]
        A = A.reshape(-1, 16)
        rec = top[order]
        real = np.isfinite(rec[:, 3])
        c_all = np.asarray(sc["center"], np.float64).reshape(-1, 3)
        r_all = np.abs(np.asarray(sc["radius"], np.float64))
        reach = float((np.linalg.norm(c_all - org, axis=1) + r_all).max())
        scale = (C.c_float * 4)()
        pair = C.c_uint32()
        assert L.mrt_debug_mfma_scale(reach, scale, C.byref(pair)) == 0
        s_ds, s_2k2, s_nk2slack, o2_max = (np.float32(v) for v in scale)
        neg_k2 = np.float32(np.array([(pair.value & 0xFFFF) << 16], np.uint32).view(np.float32)[0])
        # rays: origins on and around the spheres (within the admitted 4 x reach), unit directions
        n = 400
        k = rng.integers(0, len(sc), n)
        u = rng.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
        o = c_all[k] + u * r_all[k, None] * rng.choice([1.0, 1.0, 3.0, 0.5], n)[:, None]
        o[: n // 8] = org + rng.normal(size=(n // 8, 3)) * reach          # some far out (but admitted)
        d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
        o32, d32 = o.astype(np.float32), d.astype(np.float32)
        o_rel = (o32 - org.astype(np.float32)).astype(np.float32)
        assert ((o_rel.astype(np.float64) ** 2).sum(1) <= float(o2_max)).all()
        dsk = (d32 * s_ds).astype(np.float32)
        nk0 = (-(o_rel * dsk).sum(1)).astype(np.float32)
        k1p = ((o_rel * o_rel).sum(1).astype(np.float32) * s_nk2slack).astype(np.float32)

        def pack(v, w, tail):                                                   # the 16 K slots of a B operand, per ray
            hi = bf16_round(v)
            lo = bf16_round(v - hi)
            w0, w1, w2 = split3(w)
            t3 = np.full((len(v), 3), tail, np.float32)
            return np.concatenate([hi, lo, hi, np.stack([w0, w1, w2], 1), t3, np.zeros((len(v), 1), np.float32)], 1).astype(np.float64)
        B1 = pack(dsk, nk0, 0.0)
        B2 = pack((o_rel * s_2k2).astype(np.float32), k1p, neg_k2)
        g = A @ B1.T                                                            # [records, rays]
        assert np.abs(g[real]).max() <= 0.5
        acc = A @ B2.T + np.clip(g * np.abs(g), 0.0, 1.0)
        cand = ~(acc < 0)
        # geometry in double, from the records themselves
        oc = o[None, :, :] - rec[:, None, :3]
        b = (oc * d[None]).sum(-1)
        R2 = -rec[:, 3][:, None]
        c_infl = (oc * oc).sum(-1) - R2
        c_encl = (oc * oc).sum(-1) - R2 / (INFLATE * INFLATE)
        touches_fwd = (b * b - c_encl >= 0) & ((b < 0) | (c_encl < 0))
        must = real[:, None] & touches_fwd
        assert not (must & ~cand).any(), (name, int((must & ~cand).sum()))
        assert not cand[~real].any(), name                                      # padding records are never candidates
        dropped_on_line = real[:, None] & ~cand & (b * b - c_infl >= 0)
        R2s = np.where(real[:, None], R2, 1.0)
        behind = (b >= -1e-3 * np.sqrt(R2s)) & (c_infl > -1e-3 * R2s)           # (to the sweep's own slack)
        assert not (dropped_on_line & ~behind).any(), name
        assert (dropped_on_line.sum() > 0) and (cand[real].sum() > 0), name    # the test exercises both outcomes


def build_boxes(mrt, sc, max_levels=4, top_target=256):
    L = _lib.load()
    sc = np.ascontiguousarray(sc, mrt.SPHERE_DTYPE)
    info = (C.c_uint32 * 8)()
    assert L.mrt_debug_build_boxes(sc.ctypes.data, len(sc), max_levels, top_target, None, 0, info) == 0
    boxes = np.zeros((info[1], 8), np.float32)
    assert L.mrt_debug_build_boxes(sc.ctypes.data, len(sc), max_levels, top_target, boxes.ctypes.data, len(boxes), info) == 0
    return dict(levels=info[0], boxes=boxes, quad=bool(info[2]), base=[info[3 + k] for k in range(5)])


def test_every_box_encloses_the_spheres_under_it_and_its_slack_covers_the_discriminants_rounding(mrt):
    """The walk of large scenes tests the LINE of a ray against the axis-aligned box of the member spheres under a node, grown
    by K = kc X + kpad (kernels.hip box_may_touch, api.cpp build_boxes).  Host-side facts the conservativeness rests on:
    node j of level k covers the members [j 4^k, (j+1) 4^k); its box (f32 centre, extents measured from it) contains every
    member sphere; the boxes of a level line up with the level's records (same count, never-hit where the record is); and for
    ray origins at any distance the slack is at least 1.4143 x how far beyond a member's surface the line of a ray with a
    computed discriminant >= 0 can pass -- min(14 eps |oc|^2 / (2 r), sqrt(14 eps) |oc|), eps = 2^-24, |oc| <= |p| + |e| --
    plus the test's own rounding 4 eps |p|_1."""
    eps = 2.0 ** -24
    rng = np.random.default_rng(3)
    for name, sc in scenes(mrt):
        if len(sc) == 0:
            continue
        for max_levels, target in [(4, 256), (4, 1), (2, 8)]:
            h, b = build(mrt, sc, max_levels, target), build_boxes(mrt, sc, max_levels, target)
            assert b["levels"] == h["levels"]
            c = np.asarray(sc["center"], np.float64).reshape(-1, 3)
            r = np.abs(np.asarray(sc["radius"], np.float64))
            n_hier = h["direct_first"] if h["n_direct"] else h["n_members"]
            never0 = np.isinf(h["nodes"][:n_hier, 3])
            for k in range(1, h["levels"] + 1):
                recs = h["top"] if k == h["levels"] else h["nodes"][h["level_base"][k]:(h["level_base"][k + 1] if k + 1 < h["levels"] else len(h["nodes"]))]
                bx = b["boxes"][b["base"][k]:(b["base"][k + 1] if k < h["levels"] else len(b["boxes"]))]
                assert len(bx) == len(recs), (name, k)
                span = 4 ** k
                for j in range(len(recs)):
                    mem = [m for m in range(j * span, min(n_hier, (j + 1) * span)) if not never0[m]]
                    if not mem:
                        assert bx[j, 3] < -1e38 and bx[j, 4] < -1e38 and bx[j, 5] < -1e38 and np.isinf(recs[j, 3]), (name, k, j)
                        continue
                    idx = h["midx"][mem]
                    ctr, ext, kc, kpad = bx[j, :3].astype(np.float64), bx[j, 3:6].astype(np.float64), float(bx[j, 6]), float(bx[j, 7])
                    assert (np.abs(c[idx] - ctr) + r[idx][:, None] <= ext * (1 + 1e-12)).all(), (name, k, j)
                    assert kc > 0 and kpad > 0
                    r_min, e2, e1 = r[idx].min(), float(np.sqrt((ext * ext).sum())), float(ext.sum())
                    for dist in 10.0 ** rng.uniform(-4, 7, 6):
                        # worst case over directions of p at this distance: |p|_1 in [dist, sqrt(3) dist], |p|_2 = dist
                        oc = dist + e2
                        need = 1.4143 * min(14 * eps * oc * oc / (2 * r_min * 0.99999), np.sqrt(14 * eps / 0.99999) * oc) + 4 * eps * np.sqrt(3) * dist
                        have = kc * dist * dist + kpad if b["quad"] else kc * dist + kpad       # (|p|_1 >= |p|_2)
                        assert have >= need, (name, k, j, dist, have, need, b["quad"])


def test_lds_footprint_keeps_the_residency_the_kernels_are_built_for(mrt):
    """The persistent grid is sized from the workgroup's LDS footprint (kernels.hip, render_lds_layout): 5 workgroups per CU for
    the headline scene (C3: 5 waves per SIMD, 96 VGPRs), 4 for large scenes (C5: the wave's work stack is sized to fill exactly a
    quarter of the CU's 160 KB, after the boxes the group keeps in LDS).  A layout change that drops either fails here instead of costing throughput unnoticed."""
    L = _lib.load()
    out = (C.c_uint32 * 3)()

    def layout(sc, max_levels=4, top_target=0):
        h = build(mrt, sc, max_levels, top_target)
        assert L.mrt_debug_lds_layout(h["n_members"], len(h["nodes"]), h["levels"], len(h["top"]), out) == 0
        return h, out[0], out[1], out[2]

    h, lds, groups, cap = layout(mrt.scene_cover(1, True)[0])
    assert h["levels"] == 1 and groups == 5 and lds * 5 <= 160 * 1024, (lds, groups)
    h, lds, groups, cap = layout(mrt.scene_stress(1, 100)[0])
    assert h["levels"] == 4 and len(h["top"]) == 64
    # every byte used: the boxes of the top and of the level below it (5 x 64 x 24 B, shared by the group's waves), the stacks
    assert groups == 4 and 4 * lds == 160 * 1024 and cap >= 480, (lds, groups, cap)
    # any large scene, whatever its hierarchy: 4 groups per CU and a work stack that holds a round's pushes several times over
    for n_side, levels, target in [(36, 4, 0), (36, 1, 64), (50, 2, 8), (70, 4, 16), (100, 3, 256)]:
        h, lds, groups, cap = layout(mrt.scene_stress(2, n_side)[0], levels, target)
        assert groups == 4 and lds * 4 <= 160 * 1024 and cap >= 400, (n_side, levels, target, lds, groups, cap)


def test_the_kernels_top_down_numbering_of_the_boxes(mrt):
    """The large-scene walk addresses boxes by ONE rule -- the children of node g are 4 g + n_top .. + 3, whatever g's level
    (kernels.hip) -- over the array api.cpp's boxes_top_down lays out.  Against the level-ordered boxes of
    mrt_debug_build_boxes: every node sits where the rule puts it (top record j at j; child q of the node at level k, index j,
    at 4 g + n_top + q), every other slot is a never-hit box, the cluster level and its parents start where the kernel is
    told, and the opened-wide copy differs only in the extents of real boxes.  The array is what the kernel reads (24 bytes a box:
    centre + extents, reported here as 8 floats): the extents carry the level-ordered box's kpad (e + kpad, rounded up: api.cpp
    pack_boxes), kc is the scene's one value."""
    L = _lib.load()
    for name, sc in scenes(mrt):
        for max_levels, target in [(4, 0), (4, 16), (3, 8), (2, 64), (1, 64)]:
            b = build_boxes(mrt, sc, max_levels, target)
            info = (C.c_uint32 * 5)()
            assert L.mrt_debug_build_boxes_top_down(sc.ctypes.data, len(sc), max_levels, target, 0, None, 0, info) == 0
            levels, n_dev, n_top, cluster_first, cluster_parent_first = (int(x) for x in info)
            assert levels == b["levels"] and n_top % 32 == 0
            dev = np.zeros((n_dev, 8), np.float32)
            wide = np.zeros((n_dev, 8), np.float32)
            assert L.mrt_debug_build_boxes_top_down(sc.ctypes.data, len(sc), max_levels, target, 0, dev.ctypes.data, n_dev, info) == 0
            assert L.mrt_debug_build_boxes_top_down(sc.ctypes.data, len(sc), max_levels, target, 1, wide.ctypes.data, n_dev, info) == 0
            o = [n_top * (4 ** t - 1) // 3 for t in range(levels + 1)]
            assert n_dev == o[levels] and cluster_first == o[levels - 1] and cluster_parent_first == (o[levels - 2] if levels >= 2 else 0)
            placed = np.zeros(n_dev, bool)
            host, base = b["boxes"], b["base"]
            real_host = host[:, 3] >= 0
            kc_scene = np.float32(host[real_host][:, 6].max()) if real_host.any() else np.float32(0)
            assert (host[real_host][:, 6] == kc_scene).all(), name              # ONE kc per scene

            def device_form(h):
                """a level-ordered box as the kernel reads it"""
                if not h[3] >= 0:
                    return np.array([h[0], h[1], h[2], h[3], h[4], h[5], 0, 0], np.float32)
                e = h[3:6].astype(np.float64) + np.float64(h[7])
                up = e.astype(np.float32)
                up = np.where(up.astype(np.float64) < e, np.nextafter(up, np.float32(np.inf)), up)
                return np.array([h[0], h[1], h[2], up[0], up[1], up[2], kc_scene, 0], np.float32)
            for t in range(levels):
                k = levels - t                               # the level at depth t
                first = base[k]
                last = base[k + 1] if k < levels else len(host)
                for j in range(last - first):
                    g = o[t] + j
                    assert np.array_equal(dev[g].view(np.uint32), device_form(host[first + j]).view(np.uint32)), (name, k, j)
                    assert (dev[g][3:6] >= host[first + j][3:6] + host[first + j][7]).all() or not host[first + j][3] >= 0
                    placed[g] = True
                    if t + 1 < levels and host[first + j][3] >= 0:          # a real node: its children by the kernel's rule
                        kids = base[k - 1] + 4 * j
                        for q in range(4):
                            assert np.array_equal(dev[4 * g + n_top + q].view(np.uint32), device_form(host[kids + q]).view(np.uint32)), (name, k, j, q)
            assert (dev[~placed][:, 3:6] == np.float32(-3.0e38)).all(), name          # never-hit everywhere else
            real = dev[:, 3] >= 0
            assert np.array_equal(wide[~real].view(np.uint32), dev[~real].view(np.uint32))
            assert (wide[real][:, 3:6] == np.float32(3.0e37)).all() and np.array_equal(wide[real][:, [0, 1, 2, 6, 7]], dev[real][:, [0, 1, 2, 6, 7]])
