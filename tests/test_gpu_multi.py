"""Multi-GPU through the C ABI: several contexts in one process + mrt_gather, the RCCL variant,
native_runner --gpus/--devices, and a plain-C caller.  On the one-GPU box the contexts share device 0 -- the
band layout, the strided copies, the stream/event ordering and every index are the same as with N devices."""
import os
import subprocess
import sys

import numpy as np
import pytest

from common import gpu_render, mismatch_report

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "myraytracer_amd", "lib")


def _sharded_states(mrt, sc, cam, args, seed, world, frames=1):
    states = []
    for r in range(world):
        st = mrt.State(args, seed=seed, shard=(r, world))
        st.set_world(sc)
        if cam is not None:
            st.set_camera(cam)
        states.append(st)
    for _ in range(frames):
        for st in states:
            st.redraw()                       # asynchronous: no sync between the shards or before the gather
    return states


@pytest.mark.parametrize("world,root,w,h", [(2, 0, 96, 54), (2, 1, 96, 54), (3, 2, 70, 45), (8, 0, 64, 200), (5, 0, 33, 7)])
def test_gather_equals_unsharded(mrt, world, root, w, h):
    """N shards rendered by N contexts and assembled by mrt_gather == the frame one context renders."""
    sc, cam = mrt.scene_cover(1, True)
    args = mrt.Args(w, h, 3, 50)
    ref, _, _ = gpu_render(mrt, sc, cam, w, h, 3, 50, 5, frames=2)
    states = _sharded_states(mrt, sc, cam, args, 5, world, frames=2)
    try:
        mrt.gather(states, root)
        got = states[root].read_gathered()
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
        assert states[root].gathered_device_ptr() != 0
        # a second frame + gather reuses the buffers and still agrees
        for st in states:
            st.redraw()
        mrt.gather(states, root)
        ref3, _, _ = gpu_render(mrt, sc, cam, w, h, 3, 50, 5, frames=3)
        got3 = states[root].read_gathered()
        assert np.array_equal(got3.view(np.uint32), ref3.view(np.uint32)), mismatch_report(got3, ref3)
    finally:
        for st in states:
            st.close()


def test_gather_per_band_copies_on_one_device(mrt):
    """The cross-device form of mrt_gather's copies (one hipMemcpyPeerAsync per band: what runs between two GPUs) forced
    onto one device, so that its band indexing is exercised on a one-GPU box: same image, uneven band counts included."""
    sc, cam = mrt.scene_cover(1, True)
    for world, root, w, h in ((3, 1, 70, 45), (8, 0, 64, 200), (5, 4, 33, 7)):
        ref, _, _ = gpu_render(mrt, sc, cam, w, h, 2, 50, 9)
        states = _sharded_states(mrt, sc, cam, mrt.Args(w, h, 2, 50), 9, world)
        try:
            states[root].debug_set_gather_per_band(True)
            mrt.gather(states, root)
            got = states[root].read_gathered()
            assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)
        finally:
            for st in states:
                st.close()


def test_gather_does_not_overtake_a_reader_on_the_roots_stream(mrt):
    """Write-after-read on the root's full frame: an asynchronous consumer of mrt_gathered_device_ptr queued on the root's
    stream (here: a long sleep kernel, then a copy) must see frame k although frame k+1 is rendered and gathered right
    behind it -- the shards' copies run on the shards' own streams and wait for an event of the root's stream."""
    import torch
    from myraytracer_amd import dist as mdist
    sc, cam = mrt.scene_cover(1, True)
    w, h, world, root = 128, 72, 3, 1
    ref1, _, _ = gpu_render(mrt, sc, cam, w, h, 2, 50, 4, frames=1)
    ref2, _, _ = gpu_render(mrt, sc, cam, w, h, 2, 50, 4, frames=2)
    assert not np.array_equal(ref1, ref2)
    stream = torch.cuda.Stream()
    states = []
    try:
        with torch.cuda.stream(stream):
            for r in range(world):
                st = mrt.State(mrt.Args(w, h, 2, 50), seed=4, shard=(r, world), stream=stream.cuda_stream if r == root else None)
                st.set_world(sc)
                st.set_camera(cam)
                states.append(st)
            for per_band in (False, True):
                states[root].debug_set_gather_per_band(per_band)
                for st in states:
                    st.reset()
                    st.redraw()
                mrt.gather(states, root)
                view = mdist.gathered_tensor(states[root])
                torch.cuda._sleep(400_000_000)              # ~0.2 s on the root's stream, then the consumer
                snap = view.clone()
                for st in states:
                    st.redraw()                             # frame k+1 on every shard, gathered at once
                mrt.gather(states, root)
                stream.synchronize()
                got1 = snap.cpu().numpy()
                got2 = states[root].read_gathered()
                assert np.array_equal(got1.view(np.uint32), ref1.view(np.uint32)), "the reader saw bands of the next frame"
                assert np.array_equal(got2.view(np.uint32), ref2.view(np.uint32))
    finally:
        for st in states:
            st.close()


def test_gather_validation(mrt):
    sc = mrt.scene_default()
    args = mrt.Args(32, 24, 1, 4)
    a = mrt.State(args, seed=1, shard=(0, 2))
    b = mrt.State(args, seed=1, shard=(0, 2))          # should be shard 1
    c = mrt.State(mrt.Args(40, 24, 1, 4), seed=1, shard=(1, 2))
    try:
        for st in (a, b, c):
            st.set_world(sc)
            st.redraw()
        with pytest.raises(mrt.MrtError) as e:
            mrt.gather([a, b], 0)
        assert e.value.status == 7 and "shard" in str(e.value)
        with pytest.raises(mrt.MrtError) as e:
            mrt.gather([a, c], 0)                       # different image size
        assert e.value.status == 7
        with pytest.raises(mrt.MrtError):
            mrt.gather([a, b], 2)                       # root out of range
        with pytest.raises(mrt.MrtError) as e:
            a.read_gathered()                           # nothing gathered yet
        assert e.value.status == 7
    finally:
        for st in (a, b, c):
            st.close()


def test_gather_rccl_on_a_caller_communicator():
    """mrt_gather_rccl with an ncclComm_t created by the caller's own librccl, in a fresh process."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_gather_check.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def _read_pfm(path, w, h):
    raw = open(path, "rb").read()
    head = f"PF\n{w} {h}\n-1.0\n".encode()
    assert raw.startswith(head)
    return np.frombuffer(raw[len(head):], np.float32).reshape(h, w, 3)


def test_native_runner_gpus(mrt, tmp_path):
    """native_runner --devices 0,0,0 (three shards, one mrt_gather per frame) writes the same image as one device."""
    exe = os.path.join(LIBDIR, "native_runner")
    common = ["--width", "120", "--height", "68", "--samples-per-frame", "4", "--ray-depth", "50", "--frames", "2", "--seed", "3",
              "--scene", "cover-glass"]
    one, three = str(tmp_path / "one.pfm"), str(tmp_path / "three.pfm")
    r = subprocess.run([exe] + common + ["--out", one], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe] + common + ["--devices", "0,0,0", "--out", three], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "3 GPU(s)" in r.stdout, r.stdout + r.stderr
    assert np.array_equal(_read_pfm(one, 120, 68).view(np.uint32), _read_pfm(three, 120, 68).view(np.uint32))
    # --gpus N asks for devices 0..N-1: on a one-GPU box device 1 does not exist and that is an error, not a fallback
    import torch
    if torch.cuda.device_count() == 1:
        r = subprocess.run([exe] + common + ["--gpus", "2"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 1 and "device 1" in r.stderr


def test_plain_c_caller_with_the_references_64_byte_world(mrt, tmp_path):
    """tests/abi_c_caller.c: the reference's raw::World (64 bytes, ending at a page boundary) + its three arrays
    through mrt_set_world_raw from C -> the committed golden image of the shipped scene."""
    import json
    from make_golden_cases import GOLDEN
    from test_abi import build_c_caller
    case = [c for c in json.load(open(os.path.join(GOLDEN, "golden.json")))["cases"] if c["name"] == "c1_default_80x45"][0]
    out = str(tmp_path / "c.bin")
    r = subprocess.run([build_c_caller(tmp_path), "render", str(case["width"]), str(case["height"]), str(case["spp"]),
                        str(case["depth"]), str(case["seed"]), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, np.float32).reshape(case["height"], case["width"], 4)
    ref = np.fromfile(os.path.join(GOLDEN, case["file"]), np.float32).reshape(case["height"], case["width"], 4)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), mismatch_report(got, ref)


def test_reference_sized_world_through_python(mrt):
    """The same 64-byte path from ctypes: the first 64 bytes of a packed Lambertian/Metal scene == raw::World."""
    sc, cam = mrt.scene_cover(1, False)                 # no dielectrics
    w, vec4, f32, i32 = mrt.pack_world(sc)
    assert w.dielectrics.length == 0
    with mrt.State(mrt.Args(64, 40, 2, 20), seed=4) as st:
        st.set_world_raw(bytes(w)[:64], vec4, f32, i32)
        st.set_camera(cam)
        st.render(1)
        a = st.read_framebuffer()
        with pytest.raises(mrt.MrtError) as e:
            st.set_world_raw(bytes(w)[:72], vec4, f32, i32)
        assert e.value.status == 1
    b, _, _ = gpu_render(mrt, sc, cam, 64, 40, 2, 20, 4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
