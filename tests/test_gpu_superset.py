"""Direct test of the conservative sweep + walk: for batches of rays, the set of spheres that reach the
render kernel's root tests must CONTAIN every sphere whose discriminant (shader.wgsl:274-282) is not < 0 unless the sphere
lies entirely behind the ray's origin (b >= 0 and c >= 0: both roots <= 0, never a hit) -- the conservativeness claim: no
false negative at the sweep or at any level of the walk; a false negative on a sphere that is not the closest would be
invisible in an image -- and must BE CONTAINED in {discriminant not < 0}, because only members whose exact discriminant
is >= 0 are queued.  The winner (index, t) is compared with the oracle's world_hit as well.

mrt_debug_world_hit runs the render kernel itself, instantiated to take its rays from an array, so sweep variant, hierarchy
depth and data layout are the ones a frame uses.  Cases: random rays through random scenes, grazing rays (aimed at
sphere limbs), far-small spheres, large coordinates, both sweep variants, every forced hierarchy depth, 10k spheres."""
import numpy as np
import pytest

from common import to_oracle_spheres

pytestmark = pytest.mark.gpu


def _normalize(O, v):
    """f32 normalize with the oracle's own arithmetic is not needed: any direction with |d.d - 1| < 1e-5 is a ray the
    kernel's sweep handles (others take its literal loop); rounding a float64 normalisation to f32 is well inside."""
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def _random_scene(mrt, rng, n, scale=1.0, offset=(0.0, 0.0, 0.0), rmin=0.05, rmax=0.6, ground=True):
    sc = np.zeros(n + (1 if ground else 0), mrt.SPHERE_DTYPE)
    sc["center"][:n] = (rng.uniform(-8, 8, (n, 3)) * [1.0, 0.15, 1.0] * scale + offset).astype(np.float32)
    sc["radius"][:n] = (rng.uniform(rmin, rmax, n) * scale).astype(np.float32)
    sc["material_ty"] = 1
    sc["albedo"] = 0.5
    if ground:
        sc["center"][n] = np.array(offset, np.float32) + np.array([0.0, -1000.0 * scale, 0.0], np.float32)
        sc["radius"][n] = 999.0 * scale
    return sc


def _rays_for(rng, sc, n_random, n_grazing, n_inside):
    """(n, 6): random rays around the scene, rays aimed at sphere limbs (|miss distance| within +-2 % and +-1e-5 of the
    radius: discriminants around 0), and rays starting on / inside spheres (secondary-bounce geometry)."""
    c, r = sc["center"].astype(np.float64), np.abs(sc["radius"].astype(np.float64))
    lo, hi = (c - r[:, None]).min(0), (c + r[:, None]).max(0)
    small = r < 50 * np.median(r)
    lo, hi = (c[small] - r[small, None]).min(0), (c[small] + r[small, None]).max(0)
    ext = np.maximum(hi - lo, 1e-3)
    out = []
    o = rng.uniform(lo - 0.5 * ext, hi + 0.5 * ext, (n_random, 3))
    d = rng.normal(size=(n_random, 3))
    out.append(np.concatenate([o, d], 1))
    if n_grazing:
        k = rng.integers(0, len(sc), n_grazing)
        o = rng.uniform(lo - 0.5 * ext, hi + 0.5 * ext, (n_grazing, 3))
        to_c = c[k] - o
        dist = np.linalg.norm(to_c, axis=1, keepdims=True)
        perp = np.cross(to_c, rng.normal(size=(n_grazing, 3)))
        perp /= np.maximum(np.linalg.norm(perp, axis=1, keepdims=True), 1e-30)
        eps = np.where(rng.random(n_grazing) < 0.5, rng.uniform(-0.02, 0.02, n_grazing), rng.uniform(-1e-5, 1e-5, n_grazing))
        limb = c[k] + perp * (r[k] * (1.0 + eps))[:, None]
        d = limb - o
        ok = (dist[:, 0] > 1.001 * r[k])
        out.append(np.concatenate([o, d], 1)[ok])
    if n_inside:
        k = rng.integers(0, len(sc), n_inside)
        u = rng.normal(size=(n_inside, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        o = c[k] + u * (r[k] * rng.choice([1.0, 0.999, 0.5, 1.001], n_inside))[:, None]
        d = rng.normal(size=(n_inside, 3))
        out.append(np.concatenate([o, d], 1))
    rays = np.concatenate(out, 0)
    rays[:, 3:] /= np.linalg.norm(rays[:, 3:], axis=1, keepdims=True)
    return rays.astype(np.float32)


def _check(mrt, O, sc, rays, sweep=0, hierarchy=None, what="", boxes=True):
    packed = O.pack_world(to_oracle_spheres(O, sc))
    ref_hit, ref_t, ref_set, required = O.world_hit_batch(packed, rays)
    with mrt.State(mrt.Args(16, 16), seed=1) as st:
        if hierarchy is not None:
            st.debug_set_hierarchy(*hierarchy)
        st.set_world(sc)
        st.debug_set_sweep(sweep)
        st.debug_set_boxes(boxes)        # large scenes: the walk's box tests (the default) or boxes opened wide (they never reject)
        variant = st.debug_sweep_variant()
        hit, t, cand = st.debug_world_hit(rays, len(sc))
    a2 = (rays[:, 3:].astype(np.float64) ** 2).sum(1)
    assert (np.abs(a2 - 1.0) < 5e-6).all()
    missing = required & ~cand
    extra = cand & ~ref_set
    assert not missing.any(), (f"{what} (sweep variant {variant}): {int(missing.sum())} (ray, sphere) pairs with discriminant >= 0 and the sphere not behind the origin never "
                               f"reached the root tests; first ray {int(np.nonzero(missing.any(1))[0][0])}")
    assert not extra.any(), f"{what}: {int(extra.sum())} pairs reached the root tests with a negative discriminant"
    assert np.array_equal(hit, ref_hit), f"{what}: winners differ on {int((hit != ref_hit).sum())} rays"
    assert np.array_equal(t.view(np.uint32)[hit >= 0], ref_t.view(np.uint32)[hit >= 0]), f"{what}: t differs"
    assert (required & ref_set).sum() == required.sum()
    return variant, int(ref_set.sum()), len(rays)


@pytest.mark.parametrize("sweep", [1, 2], ids=["valu-sweep", "matrix-core-sweep"])
def test_cover_scene_candidates_equal_the_discriminant_set(mrt, oracle, sweep):
    """C3's scene: camera-like rays from its camera position plus random, grazing and on-surface rays."""
    rng = np.random.default_rng(11)
    sc, cam = mrt.scene_cover(1, True)
    rays = _rays_for(rng, sc, 6000, 6000, 4000)
    cam_o = np.tile(np.asarray(cam.lookfrom, np.float32), (4000, 1))
    tgt = rng.uniform([-11, 0, -11], [11, 1.5, 11], (4000, 3))
    cam_d = _normalize(oracle, tgt - cam_o)
    rays = np.concatenate([rays, np.concatenate([cam_o, cam_d], 1)], 0)
    variant, pairs, n = _check(mrt, oracle, sc, rays, sweep=sweep, what="cover scene")
    assert variant == sweep and pairs > n          # more than one sphere with disc >= 0 per ray on average


@pytest.mark.parametrize("case", ["random", "far-small", "large-coordinates", "tiny", "clumps", "no-ground"])
@pytest.mark.parametrize("sweep", [1, 2], ids=["valu-sweep", "matrix-core-sweep"])
def test_random_scenes(mrt, oracle, case, sweep):
    rng = np.random.default_rng({"random": 1, "far-small": 2, "large-coordinates": 3, "tiny": 4, "clumps": 5, "no-ground": 6}[case])
    if case == "random":
        sc = _random_scene(mrt, rng, 300)
    elif case == "far-small":          # spheres of radius ~1e-3 thousands of units away from most ray origins
        sc = _random_scene(mrt, rng, 200, rmin=0.0005, rmax=0.002, ground=False)
        sc["center"] += np.array([9000.0, 0.0, 0.0], np.float32)
        near = _random_scene(mrt, rng, 100)
        sc = np.concatenate([sc, near])
    elif case == "large-coordinates":
        sc = _random_scene(mrt, rng, 300, offset=(1.0e6, -2.0e5, 3.0e5))
    elif case == "tiny":
        sc = _random_scene(mrt, rng, 300, scale=1e-2)
    elif case == "clumps":             # tight clumps far apart: the matrix-core test's slack is large here (still exact)
        sc = np.concatenate([_random_scene(mrt, rng, 60, scale=0.05, offset=o, ground=False)
                             for o in ((0, 0, 0), (1000, 0, 0), (0, 0, -1000), (-700, 300, 700))])
    else:
        sc = _random_scene(mrt, rng, 500, ground=False)
    rays = _rays_for(rng, sc, 3000, 5000, 2000)
    _check(mrt, oracle, sc, rays, sweep=sweep, what=case)


@pytest.mark.parametrize("levels,top", [(1, 1), (2, 64), (3, 16), (4, 1), (4, 256)])
def test_every_hierarchy_depth(mrt, oracle, levels, top):
    """3,000 spheres (beyond the small-scene layout) with the hierarchy forced to every depth and to a one-record top."""
    rng = np.random.default_rng(100 + levels)
    sc = _random_scene(mrt, rng, 3000, rmin=0.02, rmax=0.2)
    rays = _rays_for(rng, sc, 3000, 4000, 1000)
    for sweep in (1, 2):
        for boxes in (True, False):
            _check(mrt, oracle, sc, rays, sweep=sweep, hierarchy=(levels, top), what=f"levels {levels}, top {top}, boxes {boxes}", boxes=boxes)


def test_stress_scene_10k(mrt, oracle):
    """C5's 10,001 spheres with its automatic hierarchy: rays from its camera and grazing rays."""
    rng = np.random.default_rng(77)
    sc, cam = mrt.scene_stress(1, 100)
    rays = _rays_for(rng, sc, 1500, 2500, 500)
    cam_o = np.tile(np.asarray(cam.lookfrom, np.float32), (1500, 1))
    tgt = rng.uniform(sc["center"][:-1].min(0), sc["center"][:-1].max(0), (1500, 3))
    rays = np.concatenate([rays, np.concatenate([cam_o, _normalize(oracle, tgt - cam_o)], 1)], 0)
    _check(mrt, oracle, sc, rays, what="stress 10k")
    _check(mrt, oracle, sc, rays, what="stress 10k, boxes opened wide", boxes=False)


@pytest.mark.parametrize("spread,rmin,rmax,quad", [(2000.0, 0.01, 0.05, False), (30.0, 0.05, 0.3, True)])
def test_box_slack_forms_large_sparse_and_large_dense_scenes(mrt, oracle, spread, rmin, rmax, quad):
    """The box test's slack has two forms, chosen per scene (api.cpp build_boxes): quadratic in the origin's distance where
    the spheres are large against the scene's reach, linear where they are tiny against it (1,500 spheres of radius 0.01 - 0.05
    spread over 2,000 units: a grazing ray's discriminant error there is far more than a radius).  Both through the candidate-set
    test, grazing rays included."""
    from test_hierarchy_host import build_boxes
    rng = np.random.default_rng(int(spread))
    n = 1500
    sc = np.zeros(n, mrt.SPHERE_DTYPE)
    sc["center"] = (rng.uniform(-0.5, 0.5, (n, 3)) * [spread, 0.1 * spread, spread]).astype(np.float32)
    sc["radius"] = rng.uniform(rmin, rmax, n).astype(np.float32)
    sc["material_ty"] = 1
    sc["albedo"] = 0.5
    assert build_boxes(mrt, sc)["quad"] == quad
    rays = _rays_for(rng, sc, 3000, 6000, 1000)
    for sweep in (1, 2):
        _check(mrt, oracle, sc, rays, sweep=sweep, what=f"spread {spread}")


def test_origins_beyond_the_sweeps_scaling_take_the_literal_loop(mrt, oracle):
    """The matrix-core sweep scales its operands so that K oc.ds stays below 1/2 for origins within 4 x the scene's reach
    (api.cpp, fill_scene_params); a caller's ray from farther away must still find the reference's winner (it takes the
    index-ordered loop over all spheres, which records no candidates)."""
    rng = np.random.default_rng(5)
    sc = _random_scene(mrt, rng, 300, ground=False)
    c = sc["center"].astype(np.float64)
    k = rng.integers(0, len(sc), 3000)
    u = rng.normal(size=(3000, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    dist = rng.choice([3.0, 40.0, 200.0, 5000.0], 3000)[:, None]          # x the scene's half extent (~8): inside and far outside
    o = c[k] + u * dist * 8.0
    d = c[k] + rng.normal(size=(3000, 3)) * 0.2 - o
    rays = np.concatenate([o, d / np.linalg.norm(d, axis=1, keepdims=True)], 1).astype(np.float32)
    a2 = (rays[:, 3:].astype(np.float64) ** 2).sum(1)
    rays = rays[np.abs(a2 - 1.0) < 5e-6]
    packed = oracle.pack_world(to_oracle_spheres(oracle, sc))
    ref_hit, ref_t, _, _ = oracle.world_hit_batch(packed, rays)
    with mrt.State(mrt.Args(16, 16), seed=1) as st:
        st.set_world(sc)
        st.debug_set_sweep(2)
        hit, t, _ = st.debug_world_hit(rays, len(sc))
    assert (ref_hit >= 0).sum() > 500
    assert np.array_equal(hit, ref_hit)
    assert np.array_equal(t.view(np.uint32)[hit >= 0], ref_t.view(np.uint32)[hit >= 0])
