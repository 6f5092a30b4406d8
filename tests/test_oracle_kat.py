"""Pins the CPU oracle against everything the reference itself fixes for this path
(SURVEY.md §4 / §8c): the Xoshiro128+ integer stream, the u32->f32 conversion, the packing
indices of the shipped scene, and analytic sphere_hit / world_hit / color_sky cases.  The
reference ships no tests or golden vectors; floating-point parity with it is unpinned."""
import ctypes as C
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def f32_bits(x):
    return int(np.float32(x).view(np.uint32))


def test_xoshiro128plus_kat(oracle):
    # shader.wgsl:49-64; published xoshiro128+ vectors for state (1,2,3,4)
    s = (C.c_uint32 * 4)(1, 2, 3, 4)
    out = [oracle.lib().orc_xoshiro128plus_next(s) for _ in range(8)]
    assert out == [5, 12295, 25178119, 27286542, 39879690, 1140358681, 3276312097, 4110231701]
    assert list(s) == [857776784, 3957087773, 2428778008, 3837013768]


def test_u32_to_f32_kat(oracle):
    # shader.wgsl:66-69: f32(i)/2^32 with RNE conversion; 1.0 is reachable
    L = oracle.lib()
    for u, bits in [(0x5, 0x30A00000), (0x3007, 0x36401C00), (0x1803007, 0x3BC01804), (0x1A05C0E, 0x3BD02E07),
                    (0xFFFFFF7F, 0x3F7FFFFF), (0xFFFFFF80, 0x3F800000), (0xFFFFFFFF, 0x3F800000), (0, 0)]:
        assert f32_bits(L.orc_u32_to_f32(u)) == bits, hex(u)


def _default_scene(oracle):
    sp = np.zeros(4, oracle.SPHERE_DTYPE)   # lib.rs:687-720
    sp[0] = ((0, -100.5, -1), 100, 1, (0.8, 0.8, 0, 0))
    sp[1] = ((0, 0, -1), 0.5, 1, (0.7, 0.3, 0.3, 0))
    sp[2] = ((-1, 0, -1), 0.5, 2, (0.8, 0.8, 0.8, 0.3))
    sp[3] = ((1, 0, -1), 0.5, 2, (0.8, 0.6, 0.2, 1.0))
    return sp


def test_pack_default_scene(oracle):
    # lib.rs:722-799
    pw = oracle.pack_world(_default_scene(oracle))
    w = pw.world
    assert pw.vec4.shape == (8, 4) and len(pw.f32) == 6 and len(pw.i32) == 8
    assert (w.spheres.center_base_idx, w.spheres.radius_base_idx, w.spheres.material_ty_base_idx,
            w.spheres.material_idx_base_idx, w.spheres.length) == (0, 0, 0, 4, 4)
    assert (w.lambertians.albedo_base_idx, w.lambertians.length) == (4, 2)
    assert (w.metals.albedo_base_idx, w.metals.fuzz_base_idx, w.metals.length) == (6, 4, 2)
    assert list(pw.i32) == [1, 1, 2, 2, 0, 1, 0, 1]
    assert np.array_equal(pw.f32, np.float32([100, 0.5, 0.5, 0.5, 0.3, 1.0]))
    assert np.array_equal(pw.vec4[:, 3], np.ones(8, np.float32))
    assert np.array_equal(pw.vec4[6, :3], np.float32([0.8, 0.8, 0.8]))


def _one_sphere(oracle, center, radius):
    sp = np.zeros(1, oracle.SPHERE_DTYPE)
    sp[0] = (center, radius, 1, (0.5, 0.5, 0.5, 0))
    return oracle.pack_world(sp)


def _hit(oracle, pw, idx, orig, dir_, t_min=0.001, t_sup=1.0e4):
    h = oracle.Hit()
    r = oracle.lib().orc_sphere_hit(C.byref(pw.world), pw.vec4.ctypes.data, pw.f32.ctypes.data, pw.i32.ctypes.data,
                                    idx, (C.c_float * 3)(*orig), (C.c_float * 3)(*dir_), t_min, t_sup, C.byref(h))
    return r, h


def test_sphere_hit_analytic(oracle):
    # shader.wgsl:270-312
    pw = _one_sphere(oracle, (0, 0, -5), 1.0)
    r, h = _hit(oracle, pw, 0, (0, 0, 0), (0, 0, -1))
    assert r == 1 and h.t == 4.0 and h.front_face == 1 and list(h.normal) == [0, 0, 1] and list(h.at) == [0, 0, -4]
    # tangent ray: discriminant 0 counts as a hit (d < 0 is the miss test)
    r, h = _hit(oracle, pw, 0, (1, 0, 0), (0, 0, -1))
    assert r == 1 and h.t == 5.0
    # origin inside: near root negative -> far root, back face, flipped normal
    r, h = _hit(oracle, pw, 0, (0, 0, -5), (0, 0, -1))
    assert r == 1 and h.t == 1.0 and h.front_face == 0 and list(h.normal) == [0, 0, 1]
    # t == t_sup rejected (t_sup <= t), t == t_min accepted (t < t_min)
    assert _hit(oracle, pw, 0, (0, 0, 0), (0, 0, -1), 0.001, 4.0)[0] == 0
    assert _hit(oracle, pw, 0, (0, 0, 0), (0, 0, -1), 4.0, 1e4)[0] == 1
    # the far root is tried when the near one is out of range
    r, h = _hit(oracle, pw, 0, (0, 0, 0), (0, 0, -1), 4.5, 1e4)
    assert r == 1 and h.t == 6.0 and h.front_face == 0
    # clean miss and sphere behind the origin
    assert _hit(oracle, pw, 0, (3, 0, 0), (0, 0, -1))[0] == 0
    assert _hit(oracle, pw, 0, (0, 0, 0), (0, 0, 1))[0] == 0


def test_world_hit_tie_keeps_lowest_index(oracle):
    # shader.wgsl:291-296,320-326: `t_sup <= t` rejects later equal hits
    sp = np.zeros(3, oracle.SPHERE_DTYPE)
    sp[0] = ((0, 0, -9), 1.0, 1, (0.1, 0.1, 0.1, 0))
    sp[1] = ((0, 0, -5), 1.0, 1, (0.2, 0.2, 0.2, 0))
    sp[2] = ((0, 0, -5), 1.0, 2, (0.3, 0.3, 0.3, 0.5))
    pw = oracle.pack_world(sp)
    h, which = oracle.Hit(), C.c_int32(-1)
    r = oracle.lib().orc_world_hit(C.byref(pw.world), pw.vec4.ctypes.data, pw.f32.ctypes.data, pw.i32.ctypes.data,
                                   (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(0, 0, -1), 0.001, 1e4, C.byref(h),
                                   C.byref(which))
    assert r == 1 and which.value == 1 and h.t == 4.0 and h.ty == 1 and h.idx == 1


def test_color_sky(oracle):
    # shader.wgsl:331-334
    out = (C.c_float * 3)()
    for y, want in [(-1.0, (1.0, 1.0, 1.0)), (1.0, (0.5, 0.7, 1.0)), (0.0, (0.75, 0.85, 1.0))]:
        oracle.lib().orc_color_sky(y, out)
        assert np.allclose(list(out), want, atol=1e-7)


def test_frame_weights(oracle):
    # lib.rs:300-304 and the initial 0.0 of lib.rs:424
    assert oracle.frame_weight(0, 1.0) == 0.0
    assert oracle.frame_weight(1, 1.0) == 0.5
    assert f32_bits(oracle.frame_weight(2, 1.0)) == f32_bits(np.float32(2) / np.float32(3))
    assert oracle.frame_weight(9, 0.5) == 0.5
    assert oracle.frame_shuffle(7, 0) == [0, 0, 0, 0]
    assert oracle.frame_shuffle(7, 1) != oracle.frame_shuffle(7, 2)


def test_seeds_nonzero_and_position_keyed(oracle):
    s = oracle.fill_seeds(1, 16, 8)
    assert (s.reshape(-1, 4) != 0).any(axis=1).all()
    one = (C.c_uint32 * 4)()
    oracle.lib().orc_pixel_seed(1, 5 * 16 + 3, one)
    assert list(one) == list(s[5, 3])


def test_top_row_is_sky_gradient(oracle):
    """Statistical sanity: the top row of the shipped scene only sees sky."""
    pw = oracle.pack_world(_default_scene(oracle))
    fb = oracle.render(64, 36, 8, 8, pw, oracle.pinhole_camera(), seed=3)
    top = fb[35]
    assert np.all(top[:, 3] == 1.0)
    assert np.all(top[:, 2] == 1.0)                     # blue channel of the gradient is always 1
    assert np.all((top[:, 0] > 0.5) & (top[:, 0] < 0.75))


def test_running_mean_equals_single_frame_mean(oracle):
    """lib.rs:299-304 with max_w = 1: N frames of s spp average like one running mean."""
    pw = oracle.pack_world(_default_scene(oracle))
    cam = oracle.pinhole_camera()
    seeds = oracle.fill_seeds(5, 16, 9)
    acc = np.zeros((9, 16, 4), np.float64)
    fb = np.zeros((9, 16, 4), np.float32)
    for f in range(4):
        sh = oracle.frame_shuffle(5, f)
        single = oracle.render_frame(16, 9, 2, 8, pw, cam, seeds, sh, 0.0)
        acc += single
        fb = oracle.render_frame(16, 9, 2, 8, pw, cam, seeds, sh, oracle.frame_weight(f, 1.0), fb)
    assert np.allclose(fb, acc / 4, atol=2e-6)


def test_golden_c1_lowres(oracle):
    """Committed fixture (tests/golden/make_golden.py): regression pin of the oracle itself."""
    meta = json.load(open(os.path.join(GOLDEN, "golden.json")))
    for case in meta["cases"]:
        ref = np.fromfile(os.path.join(GOLDEN, case["file"]), np.float32).reshape(case["height"], case["width"], 4)
        from make_golden_cases import render_case
        got, counters = render_case(oracle, case)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), case["name"]
        assert counters == case["counters"], case["name"]


def test_counter_mode_sample_states(oracle):
    """Extension: per-sample states depend only on (pixel frame state, sample index)."""
    import ctypes as C
    base = (C.c_uint32 * 4)(1, 2, 3, 4)
    a, b, c = (C.c_uint32 * 4)(), (C.c_uint32 * 4)(), (C.c_uint32 * 4)()
    oracle.lib().orc_sample_state.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
    oracle.lib().orc_sample_state(base, 0, a)
    oracle.lib().orc_sample_state(base, 1, b)
    oracle.lib().orc_sample_state(base, 0, c)
    assert list(a) == list(c) and list(a) != list(b) and any(a)
    # murmur3 fmix32 of (1 + 0x9E3779B9 * 1)
    z = (1 + 0x9E3779B9) & 0xFFFFFFFF
    z ^= z >> 16; z = (z * 0x85EBCA6B) & 0xFFFFFFFF; z ^= z >> 13; z = (z * 0xC2B2AE35) & 0xFFFFFFFF; z ^= z >> 16
    assert a[0] == z
