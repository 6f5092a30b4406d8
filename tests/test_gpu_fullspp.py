"""BASELINE.json's configs at their REAL sample counts: C3 1920x1080x512, C4 3840x2160x1,024, C5 (10,001 spheres)
1920x1080x4,096, depth 50.

One lane runs a pixel's whole sequential Xoshiro128+ chain (shader.wgsl:377-382), so the chain length -- spp x
path length, up to ~10^5 bounce-loop trips per pixel at C5 -- is exactly the dimension the reduced-spp tests do not
exercise; so are the pixel FIFO / tile-queue refills and the two-frames-in-flight pipeline at their real
duration.  Each test renders the full frame on one GPU and compares whole rows bit for bit with the oracle:
rows the committed fixture holds (tests/golden/fullsize_rows.npz, made by tests/golden/make_fullsize_rows.py --
the oracle needs minutes per C5 row) and, for C3 / C4, further rows computed live.  The rows' world_hit_calls
are compared with the GPU's per-pixel costs, and C3's whole-frame counters with the oracle's.
"""
import json
import os

import numpy as np
import pytest

from common import to_oracle_camera, to_oracle_spheres
from make_golden_cases import GOLDEN

pytestmark = pytest.mark.gpu

META = json.load(open(os.path.join(GOLDEN, "fullsize_rows.json")))["configs"]
ROWS = np.load(os.path.join(GOLDEN, "fullsize_rows.npz"))


def _scene(mrt, name):
    return mrt.scene_cover(1, True) if name == "cover-glass" else mrt.scene_stress(1, 100)


def _check_fixture_rows(name, frame, fb, costs, what):
    cfg = META[name]
    for rc in cfg["row_counters"][frame]:
        y = rc["row"]
        ref = ROWS[f"{name}_f{frame}_row{y}"]
        neq = (fb[y].view(np.uint32) != ref.view(np.uint32)).any(axis=-1)
        assert not neq.any(), f"{what}: row {y}: {int(neq.sum())} of {neq.size} pixels differ, first x={int(np.nonzero(neq)[0][0])}"
        if costs is not None:           # pixel cost = bounce-loop trips = world_hit calls of the pixel (depth > 0)
            assert int(costs[y].sum()) == rc["world_hit_calls"], f"{what}: row {y} world_hit_calls"


def _live_rows(O, spheres, cam, cfg, rows, frame_inputs=None):
    packed = O.pack_world(to_oracle_spheres(O, spheres))
    seeds = O.fill_seeds(cfg["seed"], cfg["width"], cfg["height"])
    out = {}
    for y in rows:
        c = O.Counters()
        fb = O.render_frame(cfg["width"], cfg["height"], cfg["spp"], cfg["depth"], packed, to_oracle_camera(O, cam), seeds,
                            rows=(y, y + 1), counters=c)
        out[y] = (fb[y], c.as_dict())
    return out


def test_c3_at_512_spp_two_pipelined_frames(mrt, oracle):
    """The benchmarked workload itself.  Frame 0 and the accumulated frame 1 (issued back to back, so their
    render kernels overlap as in bench.py) against the fixture rows; two more rows live; whole-frame counters."""
    cfg = META["c3"]
    sc, cam = _scene(mrt, cfg["scene"])
    assert len(sc) == cfg["n_spheres"]
    with mrt.State(mrt.Args(cfg["width"], cfg["height"], cfg["spp"], cfg["depth"], 1.0), seed=cfg["seed"]) as st:
        st.set_world(sc)
        st.set_camera(cam)
        st.redraw()
        st.sync()
        f0, c0, costs0 = st.read_framebuffer(), st.read_counters(), st.debug_read_pixel_costs()
        for k in ("samples", "world_hit_calls", "rng_draws"):
            assert c0[k] == cfg["frame0_counters"][k], k
        assert int(costs0.sum()) == c0["world_hit_calls"]
        _check_fixture_rows("c3", 0, f0, costs0, "C3 frame 0")
        live = _live_rows(oracle, sc, cam, cfg, [271, 777])
        for y, (row, cnt) in live.items():
            assert np.array_equal(f0[y].view(np.uint32), row.view(np.uint32)), f"C3 live row {y}"
            assert int(costs0[y].sum()) == cnt["world_hit_calls"]
        # a fresh accumulation of two frames issued without a sync in between: frame 1 starts while frame 0 drains
        st.reset()
        st.redraw()
        st.redraw()
        st.sync()
        f1 = st.read_framebuffer()
    _check_fixture_rows("c3", 1, f1, None, "C3 frame 1 (pipelined)")
    assert np.isfinite(f1).all() and (f1[..., 3] == 1.0).all()


def test_c4_at_1024_spp(mrt, oracle):
    """C4's 3840x2160 frame at 1,024 spp, whole frame on one GPU, and -- the 8-GPU path -- ranks 0 and 5 of its 8-way
    shard, whose bands must equal the same bands of the whole frame."""
    cfg = META["c4"]
    sc, cam = _scene(mrt, cfg["scene"])
    args = mrt.Args(cfg["width"], cfg["height"], cfg["spp"], cfg["depth"], 1.0)
    with mrt.State(args, seed=cfg["seed"]) as st:
        st.set_world(sc)
        st.set_camera(cam)
        st.redraw()
        st.sync()
        full, c, costs = st.read_framebuffer(), st.read_counters(), st.debug_read_pixel_costs()
    assert c["samples"] == cfg["width"] * cfg["height"] * cfg["spp"]
    _check_fixture_rows("c4", 0, full, costs, "C4")
    live = _live_rows(oracle, sc, cam, cfg, [600])
    for y, (row, cnt) in live.items():
        assert np.array_equal(full[y].view(np.uint32), row.view(np.uint32)), f"C4 live row {y}"
        assert int(costs[y].sum()) == cnt["world_hit_calls"]
    for rank in (0, 5):
        with mrt.State(args, seed=cfg["seed"], shard=(rank, 8)) as st:
            st.set_world(sc)
            st.set_camera(cam)
            st.redraw()
            st.sync()
            part = st.read_framebuffer()
        for lr in range(0, part.shape[0], 8):
            g = mrt.shard_global_row(lr, rank, 8)
            if g < cfg["height"]:
                assert np.array_equal(part[lr:lr + 8].view(np.uint32), full[g:g + 8].view(np.uint32)), (rank, g)


def test_c5_at_4096_spp(mrt):
    """C5: 10,001 spheres, 4,096 spp -- the longest per-pixel chains of any config; fixture rows only (the oracle
    needs minutes per row)."""
    cfg = META["c5"]
    sc, cam = _scene(mrt, cfg["scene"])
    assert len(sc) == cfg["n_spheres"] == 10001
    with mrt.State(mrt.Args(cfg["width"], cfg["height"], cfg["spp"], cfg["depth"], 1.0), seed=cfg["seed"]) as st:
        st.set_world(sc)
        st.set_camera(cam)
        st.redraw()
        st.sync()
        fb, c, costs = st.read_framebuffer(), st.read_counters(), st.debug_read_pixel_costs()
    assert c["samples"] == cfg["width"] * cfg["height"] * cfg["spp"]
    assert int(costs.sum()) == c["world_hit_calls"]
    _check_fixture_rows("c5", 0, fb, costs, "C5")


def test_c5_counter_mode_at_4096_spp_whole_frame_and_eighth_shares(mrt):
    """C5 in the counter-RNG mode (the extension `bench.py --config c5 --rng counter` and the 1/8 shares of an 8-GPU C5 run
    execute): 64 block layers per pixel, layers x tiles queue items, the blocks of one pixel rendered by different lanes and
    added in block order by finalize_kernel -- against oracle rows computed in the same mode (fixture `c5ctr`: the reference's
    one stream per pixel, shader.wgsl:377-382, replaced by per-sample hashed states and blockwise sums).  Then the launch shape
    of `--gpus 8`: the shards that own the fixture rows (rank = (row / 8) mod 8), whose packed bands must hold the same rows."""
    cfg = META["c5ctr"]
    assert cfg["rng_mode"] == 1
    sc, cam = _scene(mrt, cfg["scene"])
    assert len(sc) == cfg["n_spheres"] == 10001
    args = mrt.Args(cfg["width"], cfg["height"], cfg["spp"], cfg["depth"], 1.0)
    with mrt.State(args, seed=cfg["seed"]) as st:
        st.set_world(sc)
        st.set_camera(cam)
        st.set_rng_mode(1)
        st.redraw()
        st.sync()
        fb, c, costs = st.read_framebuffer(), st.read_counters(), st.debug_read_pixel_costs()
    assert c["samples"] == cfg["width"] * cfg["height"] * cfg["spp"]
    assert int(costs.sum()) == c["world_hit_calls"]
    _check_fixture_rows("c5ctr", 0, fb, costs, "C5 counter mode")
    # and it is a different image from the stream mode's (same seed): the fixture is not the stream fixture by accident
    assert not np.array_equal(ROWS["c5ctr_f0_row500"], ROWS["c5_f0_row500"])
    world = 8
    for rc in cfg["row_counters"][0]:
        y = rc["row"]
        rank, lrow = (y // 8) % world, (y // 8) // world * 8 + y % 8
        assert mrt.shard_global_row(lrow, rank, world) == y
        with mrt.State(args, seed=cfg["seed"], shard=(rank, world)) as st:
            st.set_world(sc)
            st.set_camera(cam)
            st.set_rng_mode(1)
            st.redraw()
            st.sync()
            part, pc = st.read_framebuffer(), st.debug_read_pixel_costs()
        ref = ROWS[f"c5ctr_f0_row{y}"]
        assert np.array_equal(part[lrow].view(np.uint32), ref.view(np.uint32)), f"rank {rank} of 8: global row {y}"
        assert int(pc[lrow].sum()) == rc["world_hit_calls"]
        band = slice(lrow - lrow % 8, lrow - lrow % 8 + 8)
        assert np.array_equal(part[band].view(np.uint32), fb[y - y % 8:y - y % 8 + 8].view(np.uint32)), f"rank {rank}: band of row {y}"


def test_c5_stream_mode_eighth_shares_with_eight_frames_in_flight(mrt):
    """The launch shape of `bench.py --config c5 --gpus 8` in the reference's RNG semantics: a 1/8 share of C5 is pixel-starved
    (259,200 pixels for 262,144 resident lanes, one sequential 4,096-sample chain each), so mrt_redraw runs up to sixteen such
    frames at a time, each on an eighth of the waves (width_policy.h: eight fit, eight more wait in their queues).  For the ranks that own the fixture rows 10 / 500
    / 900: frame 0 of the share against the oracle's rows (and the pixel costs against their world_hit_calls); then three
    frames issued back to back -- in flight together -- against the same three frames rendered strictly one after the other:
    scheduling must not change a bit."""
    cfg = META["c5"]
    sc, cam = _scene(mrt, cfg["scene"])
    args = mrt.Args(cfg["width"], cfg["height"], cfg["spp"], cfg["depth"], 1.0)
    world = 8
    rows = [rc["row"] for rc in cfg["row_counters"][0]]
    assert len({(y // 8) % world for y in rows}) == len(rows)            # three different ranks
    for rc in cfg["row_counters"][0]:
        y = rc["row"]
        rank, lrow = (y // 8) % world, (y // 8) // world * 8 + y % 8
        assert mrt.shard_global_row(lrow, rank, world) == y
        with mrt.State(args, seed=cfg["seed"], shard=(rank, world)) as st:
            st.set_world(sc)
            st.set_camera(cam)
            st.redraw()
            st.sync()
            part, pc = st.read_framebuffer(), st.debug_read_pixel_costs()
            assert np.array_equal(part[lrow].view(np.uint32), ROWS[f"c5_f0_row{y}"].view(np.uint32)), f"rank {rank} of 8: global row {y}"
            assert int(pc[lrow].sum()) == rc["world_hit_calls"]
            if y != rows[0]:
                continue                    # (the pipelined comparison once: 6 more share-frames)
            st.reset()
            for _ in range(3):
                st.redraw()                 # no sync in between: the frames overlap on their own streams
            st.sync()
            piped = st.read_framebuffer()
        with mrt.State(args, seed=cfg["seed"], shard=(rank, world)) as st:
            st.debug_set_frames_in_flight(1)
            st.set_world(sc)
            st.set_camera(cam)
            for _ in range(3):
                st.redraw()
                st.sync()
            serial = st.read_framebuffer()
        assert np.array_equal(piped.view(np.uint32), serial.view(np.uint32)), "frames in flight changed the accumulated image"
