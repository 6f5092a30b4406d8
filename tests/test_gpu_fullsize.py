"""BASELINE.json's full sizes, where the oracle is too slow to render the whole frame: parity on a
row sample against the oracle, plus size-independent properties (determinism, shard invariance,
accumulation algebra, analytic sky rows)."""
import numpy as np
import pytest

from common import gpu_render, mismatch_report, oracle_render, to_oracle_camera, to_oracle_spheres

pytestmark = pytest.mark.gpu


def _oracle_rows(O, spheres, cam, w, h, spp, depth, seed, rows):
    packed = O.pack_world(to_oracle_spheres(O, spheres))
    seeds = O.fill_seeds(seed, w, h)
    out = {}
    for y in rows:
        out[y] = O.render_frame(w, h, spp, depth, packed, to_oracle_camera(O, cam), seeds, rows=(y, y + 1))[y]
    return out


def test_c2_full_resolution_row_sample(mrt, oracle):
    """C2: 1200x675, 64 spp, depth 50, Lambertian+Metal cover scene; 6 full rows checked bit for bit."""
    sc, cam = mrt.scene_cover(1, False)
    got, counters, _ = gpu_render(mrt, sc, cam, 1200, 675, 64, 50, 1)
    assert counters["samples"] == 1200 * 675 * 64
    rows = [0, 101, 300, 337, 512, 674]
    ref = _oracle_rows(oracle, sc, cam, 1200, 675, 64, 50, 1, rows)
    for y in rows:
        assert np.array_equal(got[y].view(np.uint32), ref[y].view(np.uint32)), f"row {y}"
    assert np.isfinite(got).all() and (got[..., 3] == 1.0).all()


def test_c3_full_resolution_properties(mrt, oracle):
    """C3: 1920x1080, Dielectric + defocus, depth 50 -- at 8 spp per frame so the test stays short."""
    sc, cam = mrt.scene_cover(1, True)
    a, ca, _ = gpu_render(mrt, sc, cam, 1920, 1080, 8, 50, 1)
    b, cb, _ = gpu_render(mrt, sc, cam, 1920, 1080, 8, 50, 1)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))                       # deterministic pixels
    for k in ("samples", "world_hit_calls", "rng_draws"):                               # lane_slots depends on the schedule
        assert ca[k] == cb[k]
    rows = [3, 540, 1000]
    ref = _oracle_rows(oracle, sc, cam, 1920, 1080, 8, 50, 1, rows)
    for y in rows:
        assert np.array_equal(a[y].view(np.uint32), ref[y].view(np.uint32)), f"row {y}"
    # two frames of 8 spp with the running-mean weights == mean of the two single frames (lib.rs:299-304)
    two, _, _ = gpu_render(mrt, sc, cam, 1920, 1080, 8, 50, 1, frames=2)
    with mrt.State(mrt.Args(1920, 1080, 8, 50), seed=1) as st:
        st.set_world(sc)
        st.set_camera(cam)
        st.set_rng_shuffle(mrt.frame_shuffle(1, 1))
        st.redraw()
        second_alone = st.read_framebuffer()
    assert np.allclose(two, 0.5 * (a.astype(np.float64) + second_alone), atol=1e-6)
    c = to_oracle_camera(oracle, cam)
    assert c.mode == 1


def test_c4_shape_shard_invariance(mrt):
    """C4's 3840x2160 frame tile-sharded 8 ways (one rank rendered here) matches the same bands of the
    unsharded frame; 1 spp keeps it short."""
    from myraytracer_amd import dist as mdist
    sc, cam = mrt.scene_cover(1, True)
    full, _, _ = gpu_render(mrt, sc, cam, 3840, 2160, 1, 50, 1)
    for rank in (0, 5):
        part, _, _ = gpu_render(mrt, sc, cam, 3840, 2160, 1, 50, 1, shard=(rank, 8))
        for lr in range(0, part.shape[0], 8):
            g = mdist.global_row(lr, rank, 8)
            if g < 2160:
                assert np.array_equal(part[lr:lr + 8].view(np.uint32), full[g:g + 8].view(np.uint32))


def test_c5_stress_scene_row_sample(mrt, oracle):
    """C5's 10k-sphere scene at 1920x1080 (2 spp): 3 rows against the oracle."""
    sc, cam = mrt.scene_stress(1, 100)
    assert len(sc) == 10001
    got, _, _ = gpu_render(mrt, sc, cam, 1920, 1080, 2, 50, 1)
    rows = [10, 500, 900]
    ref = _oracle_rows(oracle, sc, cam, 1920, 1080, 2, 50, 1, rows)
    for y in rows:
        assert np.array_equal(got[y].view(np.uint32), ref[y].view(np.uint32)), f"row {y}"


def test_default_scene_sky_rows_are_analytic(mrt):
    """Top rows of the shipped scene see only sky: colour = mix(white, (0.5,0.7,1), 0.5*dir.y+0.5)."""
    got, _, _ = gpu_render(mrt, mrt.scene_default(), None, 400, 225, 16, 8, 1)
    top = got[224]
    assert (top[:, 2] == 1.0).all() and (top[:, 3] == 1.0).all()
    assert ((top[:, 0] > 0.5) & (top[:, 0] < 0.76)).all()
    assert abs(float(got[..., :3].mean()) - 0.4164) < 0.01


def test_hundred_thousand_spheres(mrt, oracle):
    """10 x C5's sphere count: 99,857 spheres -> 4 hierarchy levels above 25k clusters, several sweep blocks."""
    sc, cam = mrt.scene_stress(3, 316)
    assert len(sc) > 99000
    ref = oracle_render(oracle, sc, cam, 40, 24, 2, 6, 5)
    got, c, _ = gpu_render(mrt, sc, cam, 40, 24, 2, 6, 5)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    assert c["sweep_records"] > 256          # more than one sweep block at the top level
