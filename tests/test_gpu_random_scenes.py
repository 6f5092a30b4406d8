"""Randomised parity: random spheres (all three materials, negative radii, huge and tiny spheres,
cameras inside geometry), random sizes / spp / depth / seeds / frame counts -- HIP vs oracle, bit for bit."""
import numpy as np
import pytest

from common import gpu_render, mismatch_report, oracle_render

pytestmark = pytest.mark.gpu


def _random_scene(mrt, rng, n):
    sc = np.zeros(n, mrt.SPHERE_DTYPE)
    for i in range(n):
        kind = rng.integers(0, 10)
        if kind == 0:      # huge ground-like sphere
            center, radius = (rng.uniform(-5, 5), -rng.uniform(50, 2000), rng.uniform(-5, 5)), None
            radius = abs(center[1]) - rng.uniform(0.0, 1.0)
        elif kind == 1:    # tiny sphere
            center, radius = tuple(rng.uniform(-3, 3, 3)), rng.uniform(1e-3, 2e-2)
        elif kind == 2:    # hollow-glass trick: negative radius (RTIOW), normal flips
            center, radius = tuple(rng.uniform(-3, 3, 3)), -rng.uniform(0.1, 0.8)
        else:
            center, radius = tuple(rng.uniform(-4, 4, 3)), rng.uniform(0.1, 1.5)
        ty = int(rng.integers(1, 4))
        albedo = tuple(rng.uniform(0.05, 1.0, 3))
        param = float(rng.uniform(0, 1.2)) if ty == 2 else float(rng.choice([1.0, 1.33, 1.5, 2.4, 0.7]))
        sc[i] = (center, radius, ty, albedo, param)
    return sc


@pytest.mark.parametrize("case", range(24))
def test_random_scene(mrt, oracle, case):
    rng = np.random.default_rng(1000 + case)
    n = int(rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 64, 100, 257, 520]))
    sc = _random_scene(mrt, rng, n)
    if rng.random() < 0.3:
        cam = None
    else:
        lf = rng.uniform(-6, 6, 3)
        la = rng.uniform(-1, 1, 3)
        cam = mrt.Camera(1, tuple(lf), tuple(la), (0.1 * rng.normal(), 1.0, 0.1 * rng.normal()), float(rng.uniform(15, 100)),
                         float(rng.choice([0.0, 0.0, 0.5, 3.0])), float(rng.uniform(0.5, 12)))
    w, h = int(rng.integers(1, 97)), int(rng.integers(1, 61))
    spp, depth = int(rng.choice([1, 2, 3, 5, 17])), int(rng.choice([1, 2, 5, 13, 50]))
    frames = int(rng.choice([1, 1, 2, 3]))
    max_w = float(rng.choice([1.0, 1.0, 0.75, 0.0]))
    seed = int(rng.integers(0, 2 ** 62))
    cnt = oracle.Counters()
    ref = oracle_render(oracle, sc, cam, w, h, spp, depth, seed, frames, max_w, counters=cnt)
    got, c, _ = gpu_render(mrt, sc, cam, w, h, spp, depth, seed, frames, max_w)
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all(), mismatch_report(got, ref)
    assert c["samples"] == cnt.samples and c["world_hit_calls"] == cnt.world_hit_calls and c["rng_draws"] == cnt.rng_draws


@pytest.mark.parametrize("dist,radius", [(50.0, 0.01), (900.0, 0.05), (3000.0, 0.5), (9000.0, 30.0)])
def test_far_small_spheres_grazing_rays(mrt, oracle, dist, radius):
    """Where the discriminant is dominated by rounding error (|oc|^2 * eps >> r^2) the brute-force scan
    accepts and rejects spheres erratically; the clustered conservative sweep must still hand exactly the
    same candidates to the exact tests.  A dense field of tiny, far spheres seen through a narrow lens."""
    rng = np.random.default_rng(int(dist))
    n = 200
    sc = np.zeros(n + 1, mrt.SPHERE_DTYPE)
    sc[0] = ((0, -1000.0 - dist * 0.02, -dist), 1000.0, 1, (0.5, 0.5, 0.5), 0.0)
    for i in range(n):
        c = (rng.uniform(-1, 1) * dist * 0.02, rng.uniform(-1, 1) * dist * 0.012, -dist + rng.uniform(-1, 1) * dist * 0.02)
        sc[i + 1] = (c, radius * rng.uniform(0.5, 1.5), int(rng.integers(1, 4)), tuple(rng.uniform(0.2, 1, 3)), 0.3 if i % 2 else 1.5)
    cam = mrt.Camera(1, (0, 0, 0), (0, 0, -1), (0, 1, 0), 3.0, 0.0, 1.0)
    cnt = oracle.Counters()
    ref = oracle_render(oracle, sc, cam, 96, 64, 8, 8, 17, counters=cnt)
    got, c, _ = gpu_render(mrt, sc, cam, 96, 64, 8, 8, 17)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    assert c["world_hit_calls"] == cnt.world_hit_calls and c["rng_draws"] == cnt.rng_draws
    assert cnt.scatter_lambertian + cnt.scatter_metal + cnt.scatter_dielectric > 1000     # the spheres are really hit


def test_large_coordinates_keep_cluster_bounds_conservative(mrt, oracle):
    """Cluster centres are stored as f32; at |centre| ~ 1e6 one ulp (0.06) is comparable to the spheres'
    radii, so the bound must be measured from the rounded centre."""
    rng = np.random.default_rng(5)
    n = 64
    base = np.array([1.0e6, -2.0e5, -1.0e6])
    sc = np.zeros(n, mrt.SPHERE_DTYPE)
    for i in range(n):
        sc[i] = (tuple(base + rng.uniform(-2, 2, 3)), rng.uniform(0.05, 0.4), int(rng.integers(1, 4)), tuple(rng.uniform(0.2, 1, 3)), 0.2 if i % 2 else 1.5)
    cam = mrt.Camera(1, tuple(base + np.array([0.0, 0.0, 12.0])), tuple(base), (0, 1, 0), 25.0, 0.0, 12.0)
    ref = oracle_render(oracle, sc, cam, 80, 60, 6, 10, 3)
    got, _, _ = gpu_render(mrt, sc, cam, 80, 60, 6, 10, 3)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)


@pytest.mark.parametrize("n", [1100, 2600])
def test_hierarchy_depth_does_not_change_pixels(mrt, oracle, n):
    """Scenes beyond 1,024 members walk a hierarchy of bounding spheres (u32 work items, nodes in HBM);
    its depth and the size of the swept top level are tuning knobs and must never show in the image."""
    rng = np.random.default_rng(n)
    sc = _random_scene(mrt, rng, n)
    cam = mrt.Camera(1, (5.0, 3.0, 6.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 50.0, 0.3, 7.0)
    cnt = oracle.Counters()
    ref = oracle_render(oracle, sc, cam, 56, 32, 3, 9, 77, counters=cnt)
    for max_levels, top_target, boxes in [(1, 64, True), (2, 64, True), (3, 8, True), (4, 1, True), (4, 64, True), (3, 8, False), (4, 64, False)]:
        with mrt.State(mrt.Args(56, 32, 3, 9, 1.0), seed=77) as st:
            st.debug_set_hierarchy(max_levels, top_target)
            st.debug_set_boxes(boxes)         # the walk's box tests (default) or boxes opened wide (they never reject): the same image
            st.set_world(sc)
            st.set_camera(cam)
            st.render(1)
            got, c = st.read_framebuffer(), st.read_counters()
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        assert same.all(), f"hierarchy {max_levels},{top_target}: " + mismatch_report(got, ref)
        assert c["world_hit_calls"] == cnt.world_hit_calls and c["rng_draws"] == cnt.rng_draws


def test_every_queue_overflows_in_a_deep_hierarchy(mrt, oracle):
    """1,100 concentric shells: every ray is a candidate for every node of every level, so the top queue is
    refilled in several passes and every lower queue runs full."""
    n = 1100
    sc = np.zeros(n, mrt.SPHERE_DTYPE)
    for i in range(n):
        sc[i] = ((0, 0, -3), 0.5 + 0.0005 * (i % 550), 1 + (i % 3), (0.8, 0.7, 0.6), 0.2 if i % 3 == 1 else 1.5)
    ref = oracle_render(oracle, sc, None, 40, 24, 2, 6, 9)
    for max_levels, top_target in [(1, 64), (4, 1)]:
        with mrt.State(mrt.Args(40, 24, 2, 6, 1.0), seed=9) as st:
            st.debug_set_hierarchy(max_levels, top_target)
            st.set_world(sc)
            st.render(1)
            got = st.read_framebuffer()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), f"hierarchy {max_levels},{top_target}: " + mismatch_report(got, ref)


@pytest.mark.parametrize("mode", [1, 2])
def test_both_sweep_variants_give_the_same_image(mrt, oracle, mode):
    """The sweep has two variants -- SGPR-fed VALU tests and bf16-split GEMMs on the matrix cores, whose slack
    grows with |o|^2 + |C|^2 -- chosen per scene and camera.  Forced either way, near the origin or 3,000
    units away from it (where the matrix-core variant makes every record a candidate), the image is the same."""
    rng = np.random.default_rng(4242)
    for n, off in [(300, 0.0), (300, 3000.0), (1500, 0.0), (40, 2.0e5)]:
        sc = _random_scene(mrt, rng, n)
        sc["center"] += np.float32(off)
        cam = mrt.Camera(1, (5.0 + off, 3.0 + off, 6.0 + off), (off, off, off), (0.0, 1.0, 0.0), 50.0, 0.3, 7.0)
        cnt = oracle.Counters()
        ref = oracle_render(oracle, sc, cam, 48, 28, 2, 7, 31, counters=cnt)
        with mrt.State(mrt.Args(48, 28, 2, 7, 1.0), seed=31) as st:
            st.debug_set_sweep(mode)
            st.set_world(sc)
            st.set_camera(cam)
            st.render(1)
            got, c = st.read_framebuffer(), st.read_counters()
        same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
        assert same.all(), f"sweep mode {mode}, n={n}, offset {off}: " + mismatch_report(got, ref)
        assert c["world_hit_calls"] == cnt.world_hit_calls and c["rng_draws"] == cnt.rng_draws


def test_sweep_variant_choice_follows_extent_not_position(mrt, oracle):
    """The matrix-core sweep works in coordinates relative to the scene's centre, so moving a scene 10,000 units
    away keeps it (and the image stays bit-identical to the oracle); what switches to the SGPR-fed sweep is a
    scene whose extent is large against its bounds: tight clumps of small spheres 1,000 units apart."""
    sc, cam = mrt.scene_cover(1, True)
    off = np.float32(10000.0)
    far = sc.copy()
    far["center"] += off
    fcam = mrt.Camera(1, tuple(np.float32(v) + off for v in cam.lookfrom), tuple(np.float32(v) + off for v in cam.lookat),
                      cam.vup, cam.vfov_deg, cam.defocus_angle_deg, cam.focus_dist)
    ref = oracle_render(oracle, far, fcam, 64, 36, 2, 10, 5)
    with mrt.State(mrt.Args(64, 36, 2, 10, 1.0), seed=5) as st:
        st.set_world(far)
        st.set_camera(fcam)
        assert st.debug_sweep_variant() == 2
        st.render(1)
        got = st.read_framebuffer()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    rng = np.random.default_rng(9)
    sparse = np.zeros(200, mrt.SPHERE_DTYPE)
    for i in range(200):                 # 50 tight clumps of 4: bounds of radius ~0.2, 1,000 units apart
        if i % 4 == 0:
            clump = rng.uniform(-500, 500, 3)
        sparse[i] = (tuple(clump + rng.uniform(-0.1, 0.1, 3)), 0.05, 1, (0.5, 0.5, 0.5), 0.0)
    with mrt.State(mrt.Args(32, 18, 1, 4, 1.0), seed=5) as st:
        st.set_world(sparse)
        assert st.debug_sweep_variant() == 1
