"""The render kernel's hand-rolled division and square root, checked directly on the device.

`div_unscaled` / `sqrt_unscaled` (kernels.hip) replace hipcc's correctly rounded expansions of `/` and sqrtf() at every
root, hit normal and normalize (shader.wgsl:286-299, :354, :381): the same operations without the operand-scaling and
fix-up steps, which only act on extreme operands.  The frames' bit-exactness is indirect evidence;
tests/test_unscaled_forms.py argues over exact rationals on the CPU.  Here the two forms run side by side on the GPU
(mrt_debug_arith): the square root of EVERY f32 of its stated domain (1.88e9 values), and more than 10^9 quotients per call site drawn from
that site's stated operand range -- bitwise equal -- then operands just outside the ranges, for which the kernel's per-wave
guards must choose the literal forms, and two crafted frames that drive whole waves down those literal branches.
"""
import numpy as np
import pytest

from common import mismatch_report, to_oracle_camera, to_oracle_spheres

pytestmark = pytest.mark.gpu


def bits(x):
    return int(np.float32(x).view(np.uint32))


@pytest.fixture(scope="module")
def ctx(mrt):
    with mrt.State(mrt.Args(8, 8, 1, 1), seed=1) as st:
        yield st


def test_sqrt_unscaled_equals_sqrtf_for_every_f32_of_its_domain(ctx):
    """x in [2^-96, FLT_MAX]: every one of the 1.88e9 values (sqrtf() rescales only below 2^-96; the kernel's call sites: disc >= 2^-96 or a
    root outside [0.001, 1e4) whatever its last bits, |ball|^2 in {0} + [2^-48, 3], 1 - x for x in [1/2, 2], [2^-60, 2^60))."""
    lo, hi = bits(2.0 ** -96), bits(np.finfo(np.float32).max)
    tested, bad, first = ctx.debug_arith(0, [lo, hi])
    assert tested == hi - lo + 1 == 1879048192
    assert bad == 0, f"{bad} square roots differ, smallest x = {np.uint32(first).view(np.float32)!r} (bits {first:#x})"


N_PAIRS = 1 << 30       # 1.07e9 per range


@pytest.mark.parametrize("site,n_lo,n_hi,d_lo,d_hi,mode", [
    # roots of sphere_hit (:290-292): numerator -b -+ sqrt(disc) of either sign, 2^-102 .. 2^95; a = dot(dir, dir) within 1e-5 of 1
    ("roots by a", 2.0 ** -102, 2.0 ** 95, 0.99999, 1.00001, 1),
    # hit normal (:299): components of at - centre, 2^-90 .. 2^30; radius of either sign, 2^-30 .. 2^30 (ABI: |v| <= 1e7 < 2^24)
    ("normal by radius", 2.0 ** -90, 2.0 ** 30, 2.0 ** -30, 2.0 ** 30, 2),
    # normalize (:354, :381): components 2^-90 .. 2^30; length = sqrt of [2^-60, 2^60)
    ("normalize by length", 2.0 ** -90, 2.0 ** 30, 2.0 ** -30, 2.0 ** 30, 1),
    # unit_sphere (:93): components of 2 rand - 1: 0 or 2^-24 .. 1; |ball| in [2^-24, 1.74]
    ("unit sphere by |ball|", 2.0 ** -24, 1.0, 2.0 ** -24, 1.75, 1),
    # moderate operands of either sign, any exponent difference up to +-90
    # (quotient exponents stay inside the window (-126, 96) in which v_div_scale passes both operands through)
    ("wide", 2.0 ** -60, 2.0 ** 60, 2.0 ** -30, 2.0 ** 30, 2),
])
def test_div_unscaled_equals_ieee_division_over_each_call_sites_range(ctx, site, n_lo, n_hi, d_lo, d_hi, mode):
    tested, bad, first = ctx.debug_arith(mode, [bits(n_lo), bits(n_hi), bits(d_lo), bits(d_hi)], count=N_PAIRS, seed=0xC0FFEE)
    assert tested == N_PAIRS
    n = np.uint32(first & 0xFFFFFFFF).view(np.float32)
    d = np.uint32(first >> 32).view(np.float32)
    assert bad == 0, f"{site}: {bad} of {tested} quotients differ, e.g. {n!r} / {d!r}"


def test_special_operands_and_the_guards_outside_the_ranges(ctx):
    """+0 and the exact cases; then operands just outside the tested sites' ranges: the kernel's per-wave predicates
    (normal_unscaled_ok, normalize_unscaled_ok -- the same device functions the render kernel calls) must reject them, which
    sends the wave down the literal `/` and sqrtf()."""
    f = np.float32
    # (x, y): inside every range -> unscaled == literal, guards accept
    inside = [(0.5, 0.5), (-0.25, 3.0), (2.0 ** -90, 2.0 ** -30), (-(2.0 ** 30), -(2.0 ** 30)), (1e-20, 1e7), (3.0, 1.75)]
    x = np.array([p[0] for p in inside], f)
    y = np.array([p[1] for p in inside], f)
    o = ctx.debug_arith_pairs(x, y)
    assert np.array_equal(o[:, 0], o[:, 1]), "quotients inside the range"
    assert (o[:, 4] == 1).all(), "hit-normal guard must accept in-range operands"
    # sqrt: +0 -> +0, as sqrtf; exact squares; the domain's ends
    xs = np.array([0.0, 1.0, 4.0, 2.0 ** -96, np.finfo(f).max, 2.0 ** -48, 3.0], f)
    o = ctx.debug_arith_pairs(xs, np.ones_like(xs))
    assert np.array_equal(o[:, 2], o[:, 3]), "square roots"
    assert o[0, 3] == 0
    # normalize guard: x = squared length, y = smallest component
    ok = ctx.debug_arith_pairs(np.array([1.0, 2.0 ** -60, 2.0 ** 59.5], f), np.array([0.5, 2.0 ** -30, 2.0 ** 29], f))
    assert (ok[:, 5] == 1).all()
    # ---- outside: a zero / tiny component, a tiny radius (hit normal); a zero / tiny component, squared length out of range (normalize)
    rel = np.array([0.0, -0.0, 2.0 ** -91, 1e-38, 1e-45, 0.5, 0.5, 0.5], f)
    rad = np.array([1.0, 1.0, 1.0, 1.0, 1.0, 2.0 ** -31, -1e-10, 0.0], f)
    o = ctx.debug_arith_pairs(rel, rad)
    assert (o[:, 4] == 0).all(), "hit-normal guard must reject: " + str(o[:, 4])
    dd = np.array([1.0, 1.0, 1.0, 2.0 ** -61, 2.0 ** 60, np.inf, np.nan, 0.0], f)
    comp = np.array([0.0, 2.0 ** -91, 1e-45, 2.0 ** -31, 2.0 ** 29, 1.0, 1.0, 0.0], f)
    o = ctx.debug_arith_pairs(dd, comp)
    assert (o[:, 5] == 0).all(), "normalize guard must reject: " + str(o[:, 5])
    # and the guards are needed: somewhere out there the unscaled forms do differ from `/` (subnormal quotients, huge ratios)
    far_n = np.array([1e-45, 1e-40, 3e-39, 1e38, 1e-30, 2.0 ** -120], f)
    far_d = np.array([3.0, 7.0, 1e5, 1e-3, 1e20, 2.0 ** 20], f)
    o = ctx.debug_arith_pairs(far_n, far_d)
    print("unscaled vs literal on far-out operands (differences are expected here):", (o[:, 0] != o[:, 1]).tolist())


def _seeds_with_zero_draws(O, seed, w, h, px, py):
    """The seed texture of `seed`, with texel (px, py) = (0, 0, b, 0): Xoshiro128+ then returns s0 + s3 = 0 twice
    (shader.wgsl:49-64: state (0,0,b,0) -> (0,b,b,0) -> ...), so the pixel's first sample has jitter exactly (0, 0)."""
    seeds = O.fill_seeds(seed, w, h).copy()
    seeds[py, px] = (0, 0, 0x9E3779B9, 0)
    return seeds


@pytest.mark.parametrize("radius", [0.5, 1e-10, -1e-10])
def test_frames_that_force_the_literal_division_branches(mrt, oracle, radius):
    """kernels.hip's two tested call sites fall back to the literal expressions when any lane of the wave has a component
    that is exactly 0 (or tiny), or a radius below 2^-30.  An odd-sized image puts a pixel centre on the optical axis
    ((px + 0.5) - 0.5 W = 0, fs_main :374); an uploaded seed texel whose first two draws are 0 makes that pixel's first
    camera ray exactly (0, 0, -1) before normalize (:381) -- zero components: the literal normalize -- and it hits a sphere
    centred on the axis head-on, so at - centre = (0, 0, r) -- zero components again: the literal normal (:299).  With radius
    +-1e-10 the sphere is hit only by that one ray (disc = 0) and at - centre = 0: tiny radius AND zero components.
    Whole frame bit-identical to the oracle, counters included (every lane of the affected waves takes the literal path)."""
    w = h = 33
    spp, depth, seed = 4, 12, 7
    M = mrt
    sc = M.World([M.Sphere((0.0, -100.5, -1.0), 100.0, M.Lambertian((0.8, 0.8, 0.0))),
                  M.Sphere((0.0, 0.0, -1.0), radius, M.Lambertian((0.7, 0.3, 0.3))),
                  M.Sphere((-1.0, 0.0, -1.0), 0.5, M.Metal((0.8, 0.8, 0.8), 0.3)),
                  M.Sphere((1.0, 0.0, -1.0), 0.5, M.Dielectric(1.5))]).to_array()
    O = oracle
    seeds = _seeds_with_zero_draws(O, seed, w, h, w // 2, h // 2)
    with M.State(M.Args(w, h, spp, depth, 1.0), seed=seed) as st:
        st.set_world(sc)
        st.set_seeds(seeds)
        st.redraw()
        st.sync()
        got, cnt = st.read_framebuffer(), st.read_counters()
    packed = O.pack_world(to_oracle_spheres(O, sc))
    c = O.Counters()
    ref = O.render_frame(w, h, spp, depth, packed, to_oracle_camera(O, None), seeds, counters=c)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    d = c.as_dict()
    for k in ("samples", "world_hit_calls", "rng_draws"):
        assert cnt[k] == d[k], k
    # the crafted ray did what the docstring says: the oracle's hit for direction (0, 0, -1) is sphere 1 head-on
    rays = np.array([[0, 0, 0, 0, 0, -1]], np.float32)
    hit, t, _, _ = O.world_hit_batch(packed, rays)
    assert int(hit[0]) == 1 and abs(float(t[0]) - (1.0 - abs(radius))) < 1e-6, (hit, t)
