"""HIP path vs CPU oracle on the same seeded inputs, through the C ABI.  Bit-exact: the
oracle and the kernels implement the same MRT-F32 rules independently, so
every framebuffer word must match; north_star's tolerance (RMSE < 1e-4) is asserted too."""
import numpy as np
import pytest

from common import gpu_render, mismatch_report, oracle_render, rmse_rgb, to_oracle_camera, to_oracle_spheres

pytestmark = pytest.mark.gpu


def _check(M, O, spheres, cam, w, h, spp, depth, seed, frames=1, max_w=1.0):
    cnt = O.Counters()
    ref = oracle_render(O, spheres, cam, w, h, spp, depth, seed, frames, max_w, counters=cnt)
    got, gcnt, _ = gpu_render(M, spheres, cam, w, h, spp, depth, seed, frames, max_w)
    assert got.shape == ref.shape
    assert rmse_rgb(got, ref) < 1e-4, mismatch_report(got, ref)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    assert gcnt["samples"] == cnt.samples
    assert gcnt["world_hit_calls"] == cnt.world_hit_calls
    assert gcnt["rng_draws"] == cnt.rng_draws


def test_seed_texture_matches_oracle(mrt, oracle):
    with mrt.State(mrt.Args(70, 37), seed=12345) as st:
        got = st.read_seeds()
    ref = oracle.fill_seeds(12345, 70, 37)
    assert np.array_equal(got[:37], ref)
    assert not got[37:].any()


def test_c1_default_scene(mrt, oracle):
    """BASELINE config C1: shipped 4-sphere scene, 400x225, 16 spp, depth 8, seed 1."""
    _check(mrt, oracle, mrt.scene_default(), None, 400, 225, 16, 8, 1)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 5), (33, 9), (64, 8), (100, 75)])
def test_ragged_sizes(mrt, oracle, w, h):
    _check(mrt, oracle, mrt.scene_default(), None, w, h, 4, 8, 3)


def test_depth_and_spp_edges(mrt, oracle):
    sc = mrt.scene_default()
    _check(mrt, oracle, sc, None, 40, 24, 3, 1, 5)      # depth 1: every hit path is exhausted
    _check(mrt, oracle, sc, None, 40, 24, 2, 0, 5)      # depth 0: colour 0, jitter draws still consumed
    _check(mrt, oracle, sc, None, 40, 24, 1, 50, 5)


def test_progressive_accumulation(mrt, oracle):
    """lib.rs:299-306: running mean over frames, and EMA when max_framebuffer_weight < 1."""
    sc = mrt.scene_default()
    _check(mrt, oracle, sc, None, 64, 36, 2, 8, 7, frames=5)
    _check(mrt, oracle, sc, None, 64, 36, 1, 8, 7, frames=6, max_w=0.5)


def test_cover_scene_metal(mrt, oracle):
    """C2's scene (Lambertian + Metal only) at reduced size."""
    sc, cam = mrt.scene_cover(1, False)
    _check(mrt, oracle, sc, cam, 120, 68, 4, 50, 1)


def test_cover_scene_dielectric_defocus(mrt, oracle):
    """C3's scene (Dielectric + defocus blur) at reduced size."""
    sc, cam = mrt.scene_cover(1, True)
    _check(mrt, oracle, sc, cam, 120, 68, 4, 50, 1)


def test_lookat_camera_reduces_to_pinhole(mrt, oracle):
    sc = mrt.scene_default()
    cam = mrt.Camera(mode=1, lookfrom=(0, 0, 0), lookat=(0, 0, -1), vup=(0, 1, 0), vfov_deg=90.0,
                     defocus_angle_deg=0.0, focus_dist=1.0)
    a, _, _ = gpu_render(mrt, sc, cam, 64, 36, 4, 8, 2)
    b, _, _ = gpu_render(mrt, sc, None, 64, 36, 4, 8, 2)
    assert rmse_rgb(a, b) < 1e-4


@pytest.mark.parametrize("scene", ["default", "cover-glass"])
def test_counter_rng_mode(mrt, oracle, scene):
    """Extension (north_star's counter-based RNG): per-sample states hashed from (pixel frame state, sample
    index).  Same parity bar: bit-identical to the oracle in that mode, and a different image from mode 0."""
    if scene == "default":
        sc, cam, args = mrt.scene_default(), None, (96, 54, 8, 8)
    else:
        sc, cam = mrt.scene_cover(1, True)
        args = (96, 54, 6, 50)
    cnt = oracle.Counters()
    ref = oracle_render(oracle, sc, cam, *args, seed=2, frames=2, counters=cnt, rng_mode=1)
    got, c, _ = gpu_render(mrt, sc, cam, *args, seed=2, frames=2, rng_mode=1)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    assert c["rng_draws"] == cnt.rng_draws and c["world_hit_calls"] == cnt.world_hit_calls
    other, _, _ = gpu_render(mrt, sc, cam, *args, seed=2, frames=2, rng_mode=0)
    assert not np.array_equal(other, got)
    assert rmse_rgb(other, got) < 0.2          # same picture, different noise


@pytest.mark.parametrize("spp", [63, 64, 65, 128, 200, 517])
def test_counter_mode_blocks_of_64_samples(mrt, oracle, spp):
    """Counter mode beyond 64 spp: a pixel's samples are summed in blocks of MRT_COUNTER_BLOCK = 64 -- possibly by different
    lanes -- and the blocks' sums are added in order; the oracle defines the same expression.  Ragged last blocks,
    exactly-full blocks, two pipelined frames, the per-pixel cost summed over blocks."""
    sc, cam = mrt.scene_cover(1, True)
    w, h, depth = 40, 24, 50
    cnt = oracle.Counters()
    ref = oracle_render(oracle, sc, cam, w, h, spp, depth, seed=6, frames=2, counters=cnt, rng_mode=1)
    with mrt.State(mrt.Args(w, h, spp, depth), seed=6) as st:
        st.set_world(sc)
        st.set_camera(cam)
        st.set_rng_mode(1)
        st.redraw()
        st.redraw()
        st.sync()
        got, c, costs = st.read_framebuffer(), st.read_counters(), st.debug_read_pixel_costs()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    assert c["samples"] == 2 * w * h * spp and c["world_hit_calls"] == cnt.world_hit_calls and c["rng_draws"] == cnt.rng_draws
    # costs are those of the LAST frame only: a single-frame oracle pass with frame 1's shuffle
    one = oracle.Counters()
    packed = oracle.pack_world(to_oracle_spheres(oracle, sc))
    oracle.render_frame(w, h, spp, depth, packed, to_oracle_camera(oracle, cam), oracle.fill_seeds(6, w, h),
                        shuffle=oracle.frame_shuffle(6, 1), counters=one, rng_mode=1)
    assert int(costs.sum()) == one.world_hit_calls


def test_counter_mode_large_scene_and_shards(mrt, oracle):
    """Counter mode with the large-scene data layout (members from L2, u32 work items: 3,000 spheres) and several
    blocks; and a 2-way shard of the same frame equals the unsharded one (blocks x bands)."""
    rng = np.random.default_rng(5)
    sc = np.zeros(3001, mrt.SPHERE_DTYPE)
    sc["center"][:3000] = (rng.uniform(-6, 6, (3000, 3)) * [1, 0.1, 1] + [0, 0.2, -8]).astype(np.float32)
    sc["radius"][:3000] = rng.uniform(0.03, 0.15, 3000).astype(np.float32)
    sc["material_ty"][:3000] = rng.choice([1, 2, 3], 3000, p=[0.7, 0.2, 0.1])
    sc["albedo"][:3000] = rng.uniform(0.2, 0.9, (3000, 3)).astype(np.float32)
    sc["param"][:3000] = np.where(sc["material_ty"][:3000] == 3, 1.5, rng.uniform(0, 0.4, 3000)).astype(np.float32)
    sc["center"][3000] = (0, -1000, -8); sc["radius"][3000] = 999.9; sc["material_ty"][3000] = 1; sc["albedo"][3000] = 0.5
    w, h, spp, depth = 48, 32, 150, 12
    ref = oracle_render(oracle, sc, None, w, h, spp, depth, seed=3, rng_mode=1)
    got, c, _ = gpu_render(mrt, sc, None, w, h, spp, depth, seed=3, rng_mode=1)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    for sweep in (1, 2):
        with mrt.State(mrt.Args(w, h, spp, depth), seed=3) as st:
            st.debug_set_sweep(sweep)
            st.set_world(sc)
            st.set_rng_mode(1)
            st.redraw()
            again = st.read_framebuffer()
        assert np.array_equal(again.view(np.uint32), ref.view(np.uint32)), f"sweep variant {sweep}"
    for rank in (0, 1):
        part, _, _ = gpu_render(mrt, sc, None, w, h, spp, depth, seed=3, rng_mode=1, shard=(rank, 2))
        for lr in range(0, part.shape[0], 8):
            g = mrt.shard_global_row(lr, rank, 2)
            if g < h:
                assert np.array_equal(part[lr:lr + 8].view(np.uint32), ref[g:g + 8].view(np.uint32)), (rank, g)


def test_every_render_kernel_instantiation_against_the_oracle(mrt, oracle):
    """render_kernel<COUNT, PILOT, CTR, SC, MFMA> has 24 frame instantiations and 12 pilot ones (the pilot never counts);
    the DBG ones (6) are what tests/test_gpu_superset.py drives.  Every one of them renders a frame here that must equal the
    oracle's: {with, without the RNG draw counter} x {stream, counter RNG} x {small-scene layout: cover scene; large-scene
    layout with the quadratic form of the box slack: 1,297 spheres; with the linear form: the same plus 100 spheres up to 5,000
    units away} x {SGPR-fed VALU sweep, matrix-core sweep}, each with the schedule forced so that the frame is preceded by a
    cost-estimating pilot launch (more tiles than persistent waves, 16 spp >= 8 x pilot spp) -- mrt_debug_last_launch reports
    which instantiations actually ran."""
    from test_hierarchy_host import build_boxes
    w, h, spp, depth = 192, 136, 16, 12
    seen_main, seen_pilot = set(), set()
    far_rng = np.random.default_rng(9)
    for layout in ("small", "large-quad", "large-linear"):
        small = layout == "small"
        sc, cam = mrt.scene_cover(1, True) if small else mrt.scene_stress(5, 36)
        if layout == "large-linear":
            far = np.zeros(100, mrt.SPHERE_DTYPE)
            far["center"] = (far_rng.uniform(-1.0, 1.0, (100, 3)) * [5000.0, 50.0, 5000.0]).astype(np.float32)
            far["radius"] = far_rng.uniform(0.1, 0.2, 100).astype(np.float32)
            far["material_ty"], far["albedo"] = 1, 0.5
            sc = np.concatenate([sc, far])
        if not small:
            assert build_boxes(mrt, sc)["quad"] == (layout == "large-quad")
        for ctr in (0, 1):
            cnt = oracle.Counters()
            ref = oracle_render(oracle, sc, cam, w, h, spp, depth, seed=12, counters=cnt, rng_mode=ctr)
            for sweep in (1, 2):
                for count in (True, False):
                    with mrt.State(mrt.Args(w, h, spp, depth), seed=12) as st:
                        st.debug_set_schedule(2, 1)                 # 1 wave per CU: fewer waves than the 408 tiles
                        st.debug_set_sweep(sweep)
                        st.set_world(sc)
                        st.set_camera(cam)
                        st.set_rng_mode(ctr)
                        st.set_draw_counting(count)
                        st.redraw()
                        got, c = st.read_framebuffer(), st.read_counters()
                        main, pilot = st.debug_last_launch()
                    what = f"{layout} ctr={ctr} sweep={sweep} count={count}"
                    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), what + ": " + mismatch_report(got, ref)
                    assert c["samples"] == cnt.samples and c["world_hit_calls"] == cnt.world_hit_calls, what
                    assert c["rng_draws"] == (cnt.rng_draws if count else 0), what
                    bits = (4 if ctr else 0) | (8 if small else 0) | (16 if sweep == 2 else 0) | (32 if layout == "large-quad" else 0)
                    assert main == (1 if count else 0) | bits, (what, main)
                    assert pilot == 2 | bits, (what, pilot)
                    seen_main.add(main)
                    seen_pilot.add(pilot)
    assert len(seen_main) == 24 and len(seen_pilot) == 12
