"""The frame loop's schedule on the GPU: forced transitions of the launch width and of the frames in flight in the middle of an
accumulation, a caller that waits for every frame, the bounded host waits and the stream-concurrency probe.

The reference has one schedule -- State::redraw draws one frame after the other (lib.rs:241-307) -- so every schedule here
must produce exactly the frames that one would: bit-identical to the serial frames and to the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from common import mismatch_report, oracle_render

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forced_schedule_transitions_mid_accumulation_are_bit_identical(mrt, oracle):
    """12 frames of the cover scene, the schedule changed under the accumulation: full width -> a quarter (4 frames in flight)
    -> an eighth (8, then 16 over-subscribed) -> a half (2) -> twice the frames at a half (4) -> back; every change waits for the frames under way and
    re-allocates slots (api.cpp, set_frame_slots).  The accumulated image must be the one 12 serial frames give, and the oracle's."""
    spheres, cam = mrt.scene_cover(1, True)
    w, h, spp, depth, frames = 192, 104, 6, 50, 12
    plan = [(1, 1), (1, 1), (4, 1), (4, 1), (8, 1), (8, 1), (8, 2), (2, 1), (2, 1), (2, 2), (2, 2), (2, 1)]
    assert len(plan) == frames
    args = mrt.Args(w, h, spp, depth, 1.0)
    with mrt.State(args, seed=11) as st:
        st.set_world(spheres); st.set_camera(cam)
        seen = []
        for div, mult in plan:
            st.set_schedule_hint(div, mult)
            st.redraw()
            sch = st.get_schedule()
            assert (sch["div"], sch["mult"], sch["settled"]) == (div, mult, True)
            # (held to the frames this process can run side by side, where the library has measured that)
            assert sch["frames_in_flight"] == min(max(2, div) * mult, sch["max_concurrent_frames"] or 16)
            seen.append(sch["frames_in_flight"])
        got = st.read_framebuffer()
        counters = st.read_counters()
    assert max(seen) >= 8 and seen[0] == 2, seen
    with mrt.State(args, seed=11) as st:                 # one frame after the other, nothing in flight
        st.set_world(spheres); st.set_camera(cam)
        st.debug_set_frames_in_flight(1)
        for _ in range(frames):
            st.redraw()
            st.sync()
        serial = st.read_framebuffer()
        serial_counters = st.read_counters()
    assert np.array_equal(got.view(np.uint32), serial.view(np.uint32)), mismatch_report(got, serial)
    for k in ("samples", "world_hit_calls", "rng_draws"):
        assert counters[k] == serial_counters[k]
    cnt = oracle.Counters()
    ref = oracle_render(oracle, spheres, cam, w, h, spp, depth, 11, frames, 1.0, counters=cnt)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
    assert counters["world_hit_calls"] == cnt.world_hit_calls and counters["rng_draws"] == cnt.rng_draws


def test_random_schedule_changes_every_few_frames_do_not_change_a_bit(mrt, oracle):
    """40 frames of a large-scene layout (boxes, work stacks) and 40 of a small one, the pinned schedule drawn anew every one to
    three frames from every valid (div, mult), with a read-back here and there (a caller that suddenly waits): each change
    drains the pipeline and re-allocates slots under the accumulation.  Against the oracle's 40 frames, bit for bit."""
    rng = np.random.default_rng(20251005)
    valid = [(d, m) for d in range(1, 9) for m in range(1, 9) if max(2, d) * m <= 16]
    for name, (spheres, cam), (w, h, spp, depth) in (("stress-36", mrt.scene_stress(3, 36), (96, 56, 3, 12)),
                                                      ("cover", mrt.scene_cover(2, True), (120, 72, 2, 20))):
        frames = 40
        with mrt.State(mrt.Args(w, h, spp, depth, 1.0), seed=77) as st:
            st.set_world(spheres); st.set_camera(cam)
            f = 0
            while f < frames:
                div, mult = valid[int(rng.integers(0, len(valid)))]
                st.set_schedule_hint(div, mult)
                for _ in range(int(rng.integers(1, 4))):
                    if f == frames:
                        break
                    st.redraw()
                    f += 1
                    if rng.random() < 0.15:
                        st.read_framebuffer()
            got = st.read_framebuffer()
            c = st.read_counters()
        cnt = oracle.Counters()
        ref = oracle_render(oracle, spheres, cam, w, h, spp, depth, 77, frames, 1.0, counters=cnt)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), name + ": " + mismatch_report(got, ref)
        assert c["world_hit_calls"] == cnt.world_hit_calls and c["rng_draws"] == cnt.rng_draws, name


def test_schedule_hint_is_validated_and_released(mrt):
    spheres, cam = mrt.scene_cover(1, True)
    with mrt.State(mrt.Args(64, 40, 4, 8, 1.0), seed=3) as st:
        st.set_world(spheres); st.set_camera(cam)
        for bad in ((0, 1), (9, 1), (8, 3), (1, 9), (3, 8)):
            with pytest.raises(mrt.MrtError):
                st.set_schedule_hint(*bad)
        st.set_schedule_hint(4, 2)
        st.redraw()
        sch = st.get_schedule()
        assert sch["frames_in_flight"] == min(8, sch["max_concurrent_frames"] or 16)
        st.set_schedule_hint(8, 2)                       # twice the launches the chip holds
        st.redraw()
        sch = st.get_schedule()
        assert sch["frames_in_flight"] == min(16, sch["max_concurrent_frames"] or 16)
        st.set_schedule_hint(0, 0)                       # back to the measured setting
        st.redraw()
        sch = st.get_schedule()
        assert sch["div"] >= 1 and sch["frames_in_flight"] >= 2


def test_a_caller_that_waits_for_every_frame_gets_the_whole_chip(mrt):
    """The setting says an eighth of the waves per launch, eight frames in flight; a caller that reads every frame back keeps
    ONE in flight and must not be run on an eighth of the chip for long (width_policy.h, width_launch_div: a launch is never
    narrower than the most frames seen in flight over the last calls, and three calls in a row that find nothing running are
    a caller that waits).  A caller that issues its frames in bursts gets the narrow launches back."""
    spheres, cam = mrt.scene_cover(1, True)
    with mrt.State(mrt.Args(640, 360, 64, 50, 1.0), seed=5) as st:
        st.set_world(spheres); st.set_camera(cam)
        st.set_schedule_hint(8, 1)
        shares = []
        for _ in range(12):
            st.redraw()
            st.read_framebuffer()
            shares.append(st.get_schedule()["last_launch_div"])
        assert shares[0] == 8 and shares[3:] == [1] * 9, shares       # (three calls that find nothing running: a caller that waits)
        shares = []
        for _ in range(32):
            st.redraw()
            shares.append(st.get_schedule()["last_launch_div"])
        st.sync()
        assert max(shares) >= 4, shares


def test_a_stalled_wait_is_a_loud_status_not_a_hang(mrt):
    """Every blocking host wait polls with a deadline (mrt_set_wait_timeout).  With a deadline far below a frame's duration the
    back-pressure of the third redraw (two frames in flight) must come back as MRT_ERR_STALLED and name itself; the context
    then stays failed and mrt_destroy returns without waiting."""
    spheres, cam = mrt.scene_cover(1, True)
    st = mrt.State(mrt.Args(1920, 1080, 64, 50, 1.0), seed=5)
    try:
        st.set_world(spheres); st.set_camera(cam)
        st.set_schedule_hint(1, 1)
        st.sync()
        st.set_wait_timeout(2e-4)
        with pytest.raises(mrt.MrtError) as ei:
            for _ in range(4):
                st.redraw()
        assert ei.value.status == 9, ei.value               # MRT_ERR_STALLED
        msg = str(ei.value)
        assert "stalled in mrt_redraw: back-pressure of slot" in msg and "frames in flight" in msg, msg
        with pytest.raises(mrt.MrtError) as ei2:
            st.sync()
        assert ei2.value.status == 9
    finally:
        st.close()                                          # must not hang
    # the device is fine: a fresh context renders
    with mrt.State(mrt.Args(64, 40, 2, 8, 1.0), seed=1) as ok:
        ok.set_world(mrt.scene_default())
        ok.redraw()
        assert np.isfinite(ok.read_framebuffer()).all()


def test_stream_concurrency_probe_sees_the_hardware_queues(mrt):
    """The package sets GPU_MAX_HW_QUEUES=20 before the first HIP call (unless the caller set it): sixteen side streams then run
    side by side, and a pixel-starved workload gets its sixteen frames in flight."""
    if os.environ.get("GPU_MAX_HW_QUEUES") not in ("20",):
        pytest.skip("GPU_MAX_HW_QUEUES was set by the caller")
    # (streams are dealt onto the hardware queues round-robin, counting every stream the process has ever made: in a
    # long-lived process two of a context's streams may share one, so "nearly all", not "all")
    with mrt.State(mrt.Args(64, 40, 2, 8, 1.0), seed=1) as st:
        assert st.debug_stream_concurrency(8) >= 6.0
        conc = st.debug_stream_concurrency(16)
    assert conc >= 12.0, conc


_FOUR_QUEUES = r"""
import os, sys
os.environ["GPU_MAX_HW_QUEUES"] = "4"
sys.path.insert(0, sys.argv[1])
import myraytracer_amd as M
sp, cam = M.scene_stress(1, 40)
with M.State(M.Args(1920, 1080, 64, 4, 1.0), seed=1, shard=(0, 8)) as st:      # a pixel-starved share of long chains
    st.set_world(sp); st.set_camera(cam)
    st.redraw(); st.sync()
    sch = st.get_schedule()
    conc = st.debug_stream_concurrency(16)
warn = M._lib.load().mrt_last_error(None).decode()
print(sch["max_concurrent_frames"], sch["frames_in_flight"], round(conc, 2), "|", warn)
"""


def test_with_the_default_four_hardware_queues_the_schedule_holds_itself_to_them():
    """A host that left HIP's default of 4 hardware queues: the library does not touch the environment (round 4's setenv is
    gone); it measures that only about four of its streams run at a time and keeps a pixel-starved shard to four frames in
    flight, with one line of warning behind mrt_last_error(NULL)."""
    out = subprocess.run([sys.executable, "-c", _FOUR_QUEUES, ROOT], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    head, warn = out.stdout.strip().split("|", 1)
    max_frames, in_flight, conc = head.split()
    assert float(conc) < 6.5, out.stdout
    assert int(max_frames) in (2, 4) and int(in_flight) <= 4, out.stdout
    assert "GPU_MAX_HW_QUEUES" in warn and "frames in flight" in warn, warn
