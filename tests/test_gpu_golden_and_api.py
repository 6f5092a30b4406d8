"""GPU tests through the C ABI: committed golden fixtures, the raw-World upload path, explicit
seed textures, sharded rendering, error behaviour, and the diagnostics that must not change
pixels (tile launch order)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from common import gpu_render, mismatch_report, oracle_render
from make_golden_cases import GOLDEN, load_inputs

pytestmark = pytest.mark.gpu


def _cases():
    return json.load(open(os.path.join(GOLDEN, "golden.json")))["cases"]


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
def test_golden_fixture(mrt, case):
    raw, cam = load_inputs(case)
    spheres = raw.view(mrt.SPHERE_DTYPE)
    camera = None if cam["mode"] == 0 else mrt.Camera(1, cam["lookfrom"], cam["lookat"], cam["vup"], cam["vfov_deg"],
                                                      cam["defocus_angle_deg"], cam["focus_dist"])
    got, counters, _ = gpu_render(mrt, spheres, camera, case["width"], case["height"], case["spp"], case["depth"],
                                  case["seed"], case["frames"], case["max_w"])
    ref = np.fromfile(os.path.join(GOLDEN, case["file"]), np.float32).reshape(case["height"], case["width"], 4)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    for k in ("samples", "world_hit_calls", "rng_draws"):
        assert counters[k] == case["counters"][k]


def test_raw_world_upload_equals_aos_upload(mrt):
    """mrt_set_world_raw takes the reference's raw::World + three arrays verbatim (lib.rs:768-863)."""
    sc, cam = mrt.scene_cover(1, True)
    w, vec4, f32, i32 = mrt.pack_world(sc)
    with mrt.State(mrt.Args(64, 40, 2, 20), seed=4) as st:
        st.set_world_raw(w, vec4, f32, i32)
        st.set_camera(cam)
        st.render(1)
        a = st.read_framebuffer()
    b, _, _ = gpu_render(mrt, sc, cam, 64, 40, 2, 20, 4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_raw_world_validation(mrt):
    sc = mrt.scene_default()
    w, vec4, f32, i32 = mrt.pack_world(sc)
    with mrt.State(mrt.Args(16, 16), seed=1) as st:
        with pytest.raises(mrt.MrtError) as e:
            st.redraw()                                         # no scene yet
        assert e.value.status == 4
        bad = type(w).from_buffer_copy(bytes(w))
        bad.metals.fuzz_base_idx = 99                           # range outside f32_data
        with pytest.raises(mrt.MrtError) as e:
            st.set_world_raw(bad, vec4, f32, i32)
        assert e.value.status == 5
        v2 = vec4.copy()
        v2[1, 0] = np.nan
        with pytest.raises(mrt.MrtError):
            st.set_world_raw(w, v2, f32, i32)
        i2 = i32.copy()
        i2[w.spheres.material_idx_base_idx + 3] = 5             # metal index out of range
        with pytest.raises(mrt.MrtError):
            st.set_world_raw(w, vec4, f32, i2)
        st.set_world_raw(w, vec4, f32, i32)                     # still usable afterwards
        st.redraw()
        assert np.isfinite(st.read_framebuffer()).all()


def test_unknown_material_type_absorbs(mrt, oracle):
    """shader.wgsl:249-251: a material type that is neither 1 nor 2 (nor the extension 3) returns black."""
    sc = mrt.scene_default()
    w, vec4, f32, i32 = mrt.pack_world(sc)
    i32 = i32.copy()
    i32[w.spheres.material_ty_base_idx + 1] = 9
    with mrt.State(mrt.Args(48, 27, 4, 8), seed=2) as st:
        st.set_world_raw(w, vec4, f32, i32)
        st.redraw()
        got = st.read_framebuffer()
    pw = oracle.pack_world(sc.view(oracle.SPHERE_DTYPE))
    pw.i32[w.spheres.material_ty_base_idx + 1] = 9
    ref = oracle.render(48, 27, 4, 8, pw, oracle.pinhole_camera(), 2)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_explicit_seed_texture(mrt, oracle):
    """The seed texture is an input (Rgba32Uint WxH, lib.rs:397-415): any caller-supplied one is honoured."""
    rng = np.random.default_rng(7)
    seeds = rng.integers(1, 2 ** 32, size=(27, 48, 4), dtype=np.uint32)
    sc = mrt.scene_default()
    with mrt.State(mrt.Args(48, 27, 3, 8), seed=0) as st:
        st.set_world(sc)
        st.set_seeds(seeds)
        assert np.array_equal(st.read_seeds()[:27], seeds)
        st.redraw()
        got = st.read_framebuffer()
    ref = oracle.render_frame(48, 27, 3, 8, oracle.pack_world(sc.view(oracle.SPHERE_DTYPE)), oracle.pinhole_camera(), seeds)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_rng_shuffle_override_and_locals(mrt, oracle):
    sc = mrt.scene_default()
    with mrt.State(mrt.Args(32, 18, 2, 8, 0.75), seed=3) as st:
        st.set_world(sc)
        L0 = st.locals
        assert (L0.shape[0], L0.shape[1], L0.samples_per_frame, L0.ray_depth) == (32, 18, 2, 8)
        assert L0.framebuffer_weight == 0.0 and list(L0.rng_shuffle) == [0, 0, 0, 0]      # lib.rs:419-426
        st.redraw()
        L1 = st.locals
        assert L1.framebuffer_weight == 0.5 and list(L1.rng_shuffle) == mrt.frame_shuffle(3, 1)
        st.set_rng_shuffle([9, 8, 7, 6])
        st.redraw()
        assert st.frames_done == 2 and st.locals.framebuffer_weight == np.float32(2) / np.float32(3)
        got = st.read_framebuffer()
    pw = oracle.pack_world(sc.view(oracle.SPHERE_DTYPE))
    seeds = oracle.fill_seeds(3, 32, 18)
    f0 = oracle.render_frame(32, 18, 2, 8, pw, oracle.pinhole_camera(), seeds, (0, 0, 0, 0), 0.0)
    f1 = oracle.render_frame(32, 18, 2, 8, pw, oracle.pinhole_camera(), seeds, (9, 8, 7, 6), 0.5, f0)
    assert np.array_equal(got.view(np.uint32), f1.view(np.uint32))


@pytest.mark.parametrize("world,height", [(2, 45), (3, 64), (8, 100)])
def test_sharded_render_equals_unsharded(mrt, world, height):
    """Pixels are keyed by global position, so any band sharding reproduces the 1-GPU image bit for bit."""
    import torch
    from myraytracer_amd import dist as mdist
    sc, cam = mrt.scene_cover(1, True)
    full, _, _ = gpu_render(mrt, sc, cam, 72, height, 2, 30, 11)
    parts = []
    for rank in range(world):
        part, _, _ = gpu_render(mrt, sc, cam, 72, height, 2, 30, 11, shard=(rank, world))
        assert part.shape == (mdist.local_rows(height, world), 72, 4)
        parts.append(torch.from_numpy(part))
    got = mdist.unshard(torch.stack(parts), height).numpy()
    assert np.array_equal(got.view(np.uint32), full.view(np.uint32))


def test_zero_copy_framebuffer_tensor(mrt):
    import torch
    from myraytracer_amd import dist as mdist
    sc = mrt.scene_default()
    with mrt.State(mrt.Args(40, 24, 2, 8), seed=1, stream=torch.cuda.current_stream().cuda_stream) as st:
        st.set_world(sc)
        st.redraw()
        t = mdist.framebuffer_tensor(st)
        torch.cuda.synchronize()
        assert t.is_cuda and tuple(t.shape) == (24, 40, 4)
        img = mdist.gather_framebuffer(t, 24)
        assert np.array_equal(img.cpu().numpy(), st.read_framebuffer())


def test_tile_launch_order_does_not_change_pixels(mrt):
    from myraytracer_amd import _lib
    sc, cam = mrt.scene_cover(1, True)
    outs = []
    for enabled in (1, 0):
        with mrt.State(mrt.Args(160, 96, 16, 50), seed=5) as st:
            _lib.load().mrt_debug_set_tile_sort(st._ctx, enabled)
            st.set_world(sc)
            st.set_camera(cam)
            st.render(2)
            outs.append(st.read_framebuffer())
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


def test_reset_restarts_accumulation(mrt):
    sc = mrt.scene_default()
    with mrt.State(mrt.Args(32, 20, 2, 8), seed=6) as st:
        st.set_world(sc)
        st.render(3)
        a3 = st.read_framebuffer()
        st.reset()
        assert st.frames_done == 0 and st.locals.framebuffer_weight == 0.0
        st.render(3)
        assert np.array_equal(st.read_framebuffer(), a3)


def test_many_spheres_blocks_and_padding(mrt, oracle):
    """More than one 512-sphere block, a count that is not a multiple of 16, and dielectrics."""
    sc, cam = mrt.scene_stress(3, 33)          # 1090 spheres
    ref = oracle_render(oracle, sc, cam, 64, 36, 2, 12, 8)
    got, _, _ = gpu_render(mrt, sc, cam, 64, 36, 2, 12, 8)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)


def test_pathological_overlapping_spheres(mrt, oracle):
    """Every ray is a candidate for every sphere (concentric shells): stresses the candidate walk and ties."""
    n = 70
    sc = np.zeros(n, mrt.SPHERE_DTYPE)
    for i in range(n):
        sc[i] = ((0, 0, -3), 0.5 + 0.01 * (i % 35), 1 + (i % 3), (0.8, 0.7, 0.6), 0.2 if i % 3 == 1 else 1.5)
    ref = oracle_render(oracle, sc, None, 48, 27, 3, 10, 9)
    got, _, _ = gpu_render(mrt, sc, None, 48, 27, 3, 10, 9)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)


def test_empty_world_is_all_sky(mrt, oracle):
    sc = np.zeros(0, mrt.SPHERE_DTYPE)
    got, c, _ = gpu_render(mrt, sc, None, 32, 18, 2, 8, 1)
    ref = oracle_render(oracle, sc, None, 32, 18, 2, 8, 1)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert c["world_hit_calls"] == 32 * 18 * 2


def test_pipelined_frames_match_stepwise_reads(mrt, oracle):
    """Two frames are in flight inside mrt_redraw (side streams); reading the framebuffer after every
    frame, after every other frame, or only at the end must give the same accumulation."""
    sc = mrt.scene_default()
    ref = oracle_render(oracle, sc, None, 64, 40, 16, 8, 21, frames=6)
    with mrt.State(mrt.Args(64, 40, 16, 8), seed=21) as st:
        st.set_world(sc)
        seen = []
        for f in range(6):
            st.redraw()
            if f % 2 == 1:
                seen.append(st.read_framebuffer())
        assert np.array_equal(seen[-1].view(np.uint32), ref.view(np.uint32))
    with mrt.State(mrt.Args(64, 40, 16, 8), seed=21) as st:
        st.set_world(sc)
        st.render(6)
        assert np.array_equal(st.read_framebuffer().view(np.uint32), ref.view(np.uint32))
        assert st.read_counters()["samples"] == 64 * 40 * 16 * 6


def test_scene_change_between_frames(mrt, oracle):
    """Uploading a new scene while frames are in flight waits for them; the accumulation continues."""
    a, cam = mrt.scene_cover(1, True)
    b, _ = mrt.scene_cover(2, True)
    with mrt.State(mrt.Args(48, 32, 16, 20), seed=3) as st:
        st.set_world(a)
        st.set_camera(cam)
        st.render(2)
        st.set_world(b)
        st.render(2)
        got = st.read_framebuffer()
    from common import to_oracle_camera, to_oracle_spheres
    seeds = oracle.fill_seeds(3, 48, 32)
    fb = np.zeros((32, 48, 4), np.float32)
    for f in range(4):
        pw = oracle.pack_world(to_oracle_spheres(oracle, a if f < 2 else b))
        fb = oracle.render_frame(48, 32, 16, 20, pw, to_oracle_camera(oracle, cam), seeds, oracle.frame_shuffle(3, f),
                                 oracle.frame_weight(f, 1.0), fb)
    assert np.array_equal(got.view(np.uint32), fb.view(np.uint32))


def test_low_spp_frames_skip_the_pilot_and_still_match(mrt, oracle):
    """samples_per_frame = 1 (the reference's default): no pilot pass, index-order queue on the first frames."""
    sc, cam = mrt.scene_cover(1, False)
    ref = oracle_render(oracle, sc, cam, 80, 48, 1, 50, 5, frames=5)
    got, _, _ = gpu_render(mrt, sc, cam, 80, 48, 1, 50, 5, frames=5)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)


def test_degenerate_rays_follow_the_reference_nan_semantics(mrt, oracle):
    """A metal with an absurd fuzz overflows `dir` so that normalize() returns the zero vector; the next
    sphere tests then have a == 0, t = -0/0 = NaN, and the reference's comparisons (`t < t_min || t_sup <= t`
    false for NaN, shader.wgsl:291-296) ACCEPT every sphere, after which the ray origin is NaN.  The kernel's
    non-finite-ray path must reproduce all of it bit for bit, RNG draw count included."""
    sc = mrt.scene_default()
    sc["param"][2] = 1e30
    sc["param"][3] = 3e38
    cnt = oracle.Counters()
    ref = oracle_render(oracle, sc, None, 64, 36, 8, 12, 4, counters=cnt)
    got, c, _ = gpu_render(mrt, sc, None, 64, 36, 8, 12, 4)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    assert c["rng_draws"] == cnt.rng_draws and c["world_hit_calls"] == cnt.world_hit_calls
    assert cnt.paths_exhausted > 1000          # the degenerate chains really happened


def test_native_runner_cli(mrt, tmp_path):
    """The headless counterpart of native-runner (main.rs:20-31): same five flags, writes an image."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "myraytracer_amd", "lib", "native_runner")
    out = str(tmp_path / "o.pfm")
    r = subprocess.run([exe, "--width", "64", "--height", "36", "--samples-per-frame", "4", "--ray-depth", "8",
                        "--max-framebuffer-weight", "1.0", "--frames", "2", "--seed", "1", "--out", out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = open(out, "rb").read()
    head = b"PF\n64 36\n-1.0\n"
    assert raw.startswith(head)
    img = np.frombuffer(raw[len(head):], np.float32).reshape(36, 64, 3)
    ref, _, _ = gpu_render(mrt, mrt.scene_default(), None, 64, 36, 4, 8, 1, frames=2)
    assert np.array_equal(img, ref[..., :3])
    r = subprocess.run([exe, "--bogus", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2


def test_zero_samples_per_frame_is_nan_like_the_reference(mrt, oracle):
    """shader.wgsl:383: color / f32(sample_count) with sample_count == 0 is 0/0; alpha stays 1."""
    got, c, _ = gpu_render(mrt, mrt.scene_default(), None, 24, 16, 0, 8, 1)
    ref = oracle_render(oracle, mrt.scene_default(), None, 24, 16, 0, 8, 1)
    assert np.isnan(got[..., :3]).all() and np.isnan(ref[..., :3]).all()
    assert (got[..., 3] == 1.0).all() and (ref[..., 3] == 1.0).all()
    assert c["samples"] == 0 and c["world_hit_calls"] == 0


@pytest.mark.parametrize("frames,max_w", [(2, 1.0), (5, 1.0), (11, 0.7), (40, 1.0)])
def test_render_batches_frames_without_changing_them(mrt, oracle, frames, max_w):
    """mrt_render(K) may render several consecutive frames with ONE launch (a small image cannot fill the GPU with one
    frame's pixels): every frame keeps the rng_shuffle, the weight and the blend order it has with K x mrt_redraw -- the
    accumulated image is bit-identical to the stepwise one and to the oracle's progressive loop."""
    sc, cam = mrt.scene_cover(1, True)
    w, h, spp, depth = 56, 40, 9, 50
    ref = oracle_render(oracle, sc, cam, w, h, spp, depth, seed=8, frames=frames, max_w=max_w)
    with mrt.State(mrt.Args(w, h, spp, depth, max_w), seed=8) as st:
        st.set_world(sc)
        st.set_camera(cam)
        st.render(frames)                       # batched: up to 32 frames per launch
        batched, cb, costs_b = st.read_framebuffer(), st.read_counters(), st.debug_read_pixel_costs()
        assert st.frames_done == frames
        kernels_batched = len(st.kernel_ms_history(64))
    with mrt.State(mrt.Args(w, h, spp, depth, max_w), seed=8) as st:
        st.set_world(sc)
        st.set_camera(cam)
        for _ in range(frames):
            st.redraw()
        stepwise, cs, costs_s = st.read_framebuffer(), st.read_counters(), st.debug_read_pixel_costs()
        kernels_stepwise = len(st.kernel_ms_history(64))
    assert np.array_equal(batched.view(np.uint32), ref.view(np.uint32)), mismatch_report(batched, ref)
    assert np.array_equal(stepwise.view(np.uint32), ref.view(np.uint32))
    for k in ("samples", "world_hit_calls", "rng_draws"):
        assert cb[k] == cs[k]
    assert np.array_equal(costs_b, costs_s)                      # the last frame's per-pixel costs
    assert kernels_stepwise == frames and kernels_batched == -(-frames // 32)      # up to 32 frames per launch


@pytest.mark.parametrize("form", [2, 3])
@pytest.mark.parametrize("w,h,spp,depth,frames,max_w", [(61, 43, 1, 50, 37, 1.0), (56, 40, 3, 50, 9, 0.8), (40, 24, 1, 0, 5, 1.0), (40, 24, 2, 1, 33, 1.0)])
def test_both_forms_of_a_frame_batch_give_the_stepwise_frames(mrt, oracle, form, w, h, spp, depth, frames, max_w):
    """A batch takes one of two forms (mrt_render): its frames as layers of the tile queue (a pixel-starved shard) or -- short
    frames of an image with pixels enough, the reference's default 1 sample per frame (lib.rs:27-37, :299-306) -- the lane
    that takes a pixel renders it for every frame of the batch, re-seeding from the seed texel and each frame's own
    rng_shuffle (shader.wgsl:44-47).  Forced onto small ragged images here: both give the oracle's progressive frames."""
    sc, cam = mrt.scene_cover(1, True)
    ref = oracle_render(oracle, sc, cam, w, h, spp, depth, seed=21, frames=frames, max_w=max_w)
    with mrt.State(mrt.Args(w, h, spp, depth, max_w), seed=21) as st:
        st.set_world(sc)
        st.set_camera(cam)
        st.debug_set_frame_batching(form)
        st.render(frames)
        got, c, costs = st.read_framebuffer(), st.read_counters(), st.debug_read_pixel_costs()
        assert st.frames_done == frames and len(st.kernel_ms_history(64)) == -(-frames // 32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    assert c["samples"] == w * h * spp * frames
    with mrt.State(mrt.Args(w, h, spp, depth, max_w), seed=21) as st:
        st.set_world(sc)
        st.set_camera(cam)
        for _ in range(frames):
            st.redraw()
        cs, costs_s = st.read_counters(), st.debug_read_pixel_costs()
    for k in ("samples", "world_hit_calls", "rng_draws"):
        assert c[k] == cs[k], k
    assert np.array_equal(costs, costs_s)                       # the last frame's per-pixel costs


def test_interactive_frames_of_a_large_image_share_launches(mrt, oracle):
    """The reference's default operating point (1 sample per frame, rendered forever: lib.rs:27-37, :187-192) at a size
    where mrt_render chooses the in-lane form by itself (as many tiles as two per persistent wave): 40 frames of 1 spp at
    1024 x 640, EMA weight 0.9, against the oracle's progressive loop; two launches."""
    sc, cam = mrt.scene_cover(1, True)
    w, h, frames = 1024, 640, 40
    ref = oracle_render(oracle, sc, cam, w, h, 1, 50, seed=2, frames=frames, max_w=0.9)
    with mrt.State(mrt.Args(w, h, 1, 50, 0.9), seed=2) as st:
        st.set_world(sc)
        st.set_camera(cam)
        st.render(frames)
        got = st.read_framebuffer()
        assert len(st.kernel_ms_history(64)) == 2
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)


def test_launches_without_the_draw_counter_render_the_same_frames(mrt, oracle):
    """mrt_set_draw_counting(0) selects the instantiation without the per-lane RNG draw counter (what bench.py times): the
    same image, samples / world_hit_calls still counted, rng_draws no longer advancing -- stream and counter RNG modes, small
    and large scene layouts."""
    for sc, cam, n_big in ((mrt.scene_cover(1, True) + (False,)), (mrt.scene_stress(2, 36) + (True,))):
        for rng_mode in (0, 1):
            w, h, spp, depth = 72, 40, 5, 30
            cnt = oracle.Counters()
            ref = oracle_render(oracle, sc, cam, w, h, spp, depth, seed=6, frames=2, counters=cnt, rng_mode=rng_mode)
            with mrt.State(mrt.Args(w, h, spp, depth, 1.0), seed=6) as st:
                st.set_world(sc)
                st.set_camera(cam)
                st.set_rng_mode(rng_mode)
                st.redraw()
                c1 = st.read_counters()
                st.set_draw_counting(False)
                st.redraw()
                got, c2 = st.read_framebuffer(), st.read_counters()
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
            assert c2["rng_draws"] == c1["rng_draws"] and 0 < c1["rng_draws"] < cnt.rng_draws
            assert c2["samples"] == cnt.samples == 2 * w * h * spp and c2["world_hit_calls"] == cnt.world_hit_calls


def test_render_batching_respects_overrides_and_switch(mrt):
    """A caller-supplied rng_shuffle applies to the next frame only, which is then rendered on its own; the A/B switch turns
    batching off; either way the images agree."""
    sc = mrt.scene_default()
    args = mrt.Args(48, 32, 5, 8)
    with mrt.State(args, seed=3) as st:
        st.set_world(sc)
        st.set_rng_shuffle([11, 22, 33, 44])
        st.render(4)
        a = st.read_framebuffer()
        n_a = len(st.kernel_ms_history(64))
    with mrt.State(args, seed=3) as st:
        st.set_world(sc)
        st.set_rng_shuffle([11, 22, 33, 44])
        for _ in range(4):
            st.redraw()
        b = st.read_framebuffer()
    with mrt.State(args, seed=3) as st:
        st.set_world(sc)
        st.debug_set_frame_batching(False)
        st.set_rng_shuffle([11, 22, 33, 44])
        st.render(4)
        c = st.read_framebuffer()
        n_c = len(st.kernel_ms_history(64))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a.view(np.uint32), c.view(np.uint32))
    assert n_a == 2 and n_c == 4              # overridden frame alone + a batch of three; four single launches
