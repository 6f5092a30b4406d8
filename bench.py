#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Started plainly with --gpus N > 1 it launches the N ranks itself (fresh child processes through
torch.distributed.run, before this process has touched the GPU) and relays rank 0's JSON line.

One "step" = one `State::redraw` (raytracer/src/lib.rs:241-307): a full raytrace pass of
samples_per_frame spp over the whole image, blended into the accumulated framebuffer, and
-- for N > 1 -- the RCCL gather of every rank's bands to rank 0.

Workload (BASELINE.json): the metric is quoted on "1920x1080 random-spheres" = configs[2]
(C3: RTIOW cover scene with Dielectric + defocus blur, 1920x1080, 512 spp, depth 50).
Multi-GPU is weak scaling towards configs[3] (C4 = 8 x C3's samples at 8 GPUs): the same scene
and camera with N x C3's samples -- N=1 1920x1080x512, N=2 2716x1528x512 (2.001 x the pixels, same
16:9 framing), N=4 3840x2160x512, N=8 3840x2160x1024 (= C4); the image is tile-sharded in
interleaved 8-row bands, every rank renders (1/N) of it.

--config c4 / c5 select BASELINE.json's configs[3] / configs[4] as they are stated (3840x2160x1024 cover
scene; 1920x1080x4096 on the 10k-sphere stress scene): a fixed frame split over the N GPUs ("strong").

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (HBM, as
north_star asks; this path is VALU-bound so `valu` carries the binding fraction) and, at
N=1, `cpu_baseline` (the oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCLK_HZ = 2.4e9            # shader clock under this load (rocm-smi while the bench runs: 2,393 MHz)
HANG_EXIT_CODE = 75       # a watchdog ended a hung N > 1 leg: the JSON line is complete and records the hang
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
LANE_OPS_PEAK = 256 * 4 * 32 * 2.4e9   # CUs x SIMDs x lanes/clk x Hz: fp32 VALU lane-ops/s
VALU_PER_BOUND_TEST = {1: 11, 2: 2}   # sweep variant 1: 10 fp32 VALU + 1 v_alignbit from SGPRs; 2: 1 fma + 1 v_alignbit after the matrix cores
VALU_PER_MEMBER_TEST = 13    # 11 fp32 VALU + compare + queue bookkeeping per member discriminant

WORKLOADS = {   # config c3 (default), weak scaling towards C4: n_gpus -> (width, height, spp)
    1: (1920, 1080, 512),
    2: (2716, 1528, 512),
    4: (3840, 2160, 512),
    8: (3840, 2160, 1024),
}
FIXED_CONFIGS = {   # BASELINE.json's other configs as stated, whatever N: (scene, width, height, spp, depth)
    "c1": ("default", 400, 225, 16, 8),          # configs[0]: the reference's shipped scene (4 spheres)
    "c2": ("cover", 1200, 675, 64, 50),          # configs[1]: cover scene, Lambertian + Metal only
    "c4": ("cover-glass", 3840, 2160, 1024, 50), # configs[3]
    "c5": ("stress", 1920, 1080, 4096, 50),      # configs[4]
    # the reference's own operating point: Args::default() renders 1 sample per frame, forever (lib.rs:27-37, :187-192,
    # :299-306; native-runner/src/main.rs:24-25) -- one State::redraw per step; with --frames-per-step K one mrt_render(K)
    "interactive": ("cover-glass", 1920, 1080, 1, 50),
}


def launch_ranks(n, argv):
    """--gpus N without a launcher: start the N ranks as fresh children (this process has not touched the GPU and
    never will), relay their output -- rank 0 prints the JSON line -- and exit with their status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES", "20"))      # frames in flight of pixel-starved shards (api.cpp)
    return subprocess.run(cmd, env=env).returncode


def host_cores():
    n = len(os.sched_getaffinity(0))
    try:   # cgroup v2 quota
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def cpu_baseline(spheres, cam, width, height, depth, seed, budget_s=15.0):
    """Time the CPU oracle (kind "port") on a bounded sample of the same workload.  The oracle scans every sphere for every ray,
    as the reference does (shader.wgsl:314-329): with thousands of spheres one 1-spp frame of the full image would take
    minutes, so the sample is then a lower-resolution frame of the same scene and camera (the rate per sample is what is
    reported; the note says what was rendered)."""
    full = (width, height)
    while len(spheres) * width * height > 1.2e9 and width >= 240:
        width, height = width // 2, height // 2
    from oracle import pyoracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import to_oracle_camera, to_oracle_spheres
    cores = min(host_cores(), O.lib().orc_max_threads())
    packed = O.pack_world(to_oracle_spheres(O, spheres))
    ocam = to_oracle_camera(O, cam)
    seeds = O.fill_seeds(seed, width, height)
    # calibrate with one full 1-spp frame, then size the timed sample to ~budget_s
    t0 = time.perf_counter()
    O.render_frame(width, height, 1, depth, packed, ocam, seeds, nthreads=cores)
    cal = time.perf_counter() - t0
    spp = max(1, min(256, int(budget_s / cal)))
    t0 = time.perf_counter()
    O.render_frame(width, height, spp, depth, packed, ocam, seeds, nthreads=cores)
    dt = time.perf_counter() - t0
    return {"value": width * height * spp / dt * 1e-6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"{width}x{height}" + (f" (the {full[0]}x{full[1]} frame at reduced resolution: {len(spheres)} spheres per ray)" if (width, height) != full else "") +
                      f", {spp} spp of the same scene/camera/depth (oracle/rt_oracle.c, "
                      f"OpenMP over rows), {dt:.1f} s; rate is spp-independent, so no extrapolation is applied"}


def source_sha16():
    """sha256 over the sources the library is built from (scripts/source_hash.py, the one definition the Makefile bakes into
    the .so as mrt_build_id() and scripts/summarize_profile.py stores next to the counters).  The GPU box has no .git."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from source_hash import source_sha16 as f
    return f(ROOT)


def abi_rccl_leg(st, dist, world, rank, steps, fence, ref_bits, in_flight):
    """N > 1, after all timing, NON-FATAL: the C ABI's own gather -- mrt_gather_rccl: grouped ncclSend / ncclRecv straight to
    the root on an ncclComm_t this caller creates with the process's librccl (torch's), then the un-permute -- instead of
    torch.distributed.gather: one verified frame, then `steps` timed redraw + gather steps.  Returns the report (rank 0) / None."""
    import ctypes as C
    path = next((l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l), None)
    if path is None:
        raise RuntimeError("no librccl mapped in this process")
    rccl = C.CDLL(path)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    in_flight["call"] = "ncclGetUniqueId / broadcast of the id"
    if rank == 0:
        rc = rccl.ncclGetUniqueId(C.byref(uid))
        if rc:
            raise RuntimeError(f"ncclGetUniqueId -> {rc}")
    box = [bytes(uid)]
    dist.broadcast_object_list(box, src=0)
    C.memmove(C.byref(uid), box[0], 128)
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    in_flight["call"] = "ncclCommInitRank"
    rc = rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
    if rc:
        raise RuntimeError(f"ncclCommInitRank -> {rc}")
    try:
        in_flight["call"] = "mrt_gather_rccl (grouped ncclSend / ncclRecv to rank 0), first frame"
        st.reset()
        fence()
        st.redraw()
        st.gather_rccl(comm.value, 0)
        fence()
        verified = None
        if rank == 0 and ref_bits is not None:
            import numpy as np
            verified = bool(np.array_equal(st.read_gathered().view(np.uint32), ref_bits))
        fence()
        in_flight["call"] = "mrt_gather_rccl, timed steps"
        t0 = time.perf_counter()
        for _ in range(steps):
            st.redraw()
            st.gather_rccl(comm.value, 0)
        fence()
        dt = time.perf_counter() - t0
    finally:
        in_flight["call"] = "ncclCommDestroy"
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)
    return {"verified": verified, "ms_per_step": dt / max(1, steps) * 1e3, "steps": steps, "librccl": path,
            "what": "mrt_redraw + mrt_gather_rccl (grouped ncclSend / ncclRecv to rank 0 + un-permute) per step, behind the C ABI"}


def abi_single_process_leg(a, n, width, height, spp, ref_rgb):
    """N > 1, rank 0, after the process group is gone, NON-FATAL: the one-process form of the C ABI's multi-GPU path --
    native_runner --gpus N: one context per GPU, mrt_gather's peer-to-peer band copies -- on the same workload; its image
    (one frame) against the unsharded frame, and its own timing of 2 frames after a warm-up."""
    import subprocess
    import tempfile
    import numpy as np
    exe = os.path.join(ROOT, "myraytracer_amd", "lib", "native_runner")
    # (MRT_BENCH_ABI_DEVICES=0,0 rehearses this leg on a box with fewer GPUs than ranks: the shards then share devices)
    devs = os.environ.get("MRT_BENCH_ABI_DEVICES")
    common = [exe, "--width", str(width), "--height", str(height), "--samples-per-frame", str(spp), "--ray-depth", str(a.depth),
              "--seed", "1", "--scene", a.scene, "--rng", a.rng] + (["--devices", devs] if devs else ["--gpus", str(n)])
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "frame.pfm")
        r = subprocess.run(common + ["--frames", "1", "--out", out], capture_output=True, text=True, timeout=120)
        if r.returncode != 0:
            raise RuntimeError((r.stderr or r.stdout).strip()[-300:])
        raw = open(out, "rb").read()
        head = f"PF\n{width} {height}\n-1.0\n".encode()
        img = np.frombuffer(raw[len(head):], np.float32).reshape(height, width, 3)
        same = bool(ref_rgb is not None and np.array_equal(img.view(np.uint32), ref_rgb))
    r = subprocess.run(common + ["--warmup", "1", "--frames", "2"], capture_output=True, text=True, timeout=120)
    if r.returncode != 0:
        raise RuntimeError((r.stderr or r.stdout).strip()[-300:])
    rate = float(r.stdout.split("Msamples/s")[0].split()[-1])
    return {"value": rate, "unit": "Msamples/s", "image_equals_unsharded_frame": same if ref_rgb is not None else None,
            "what": f"native_runner --gpus {n}: one process, one mrt_ctx per GPU, mrt_gather (peer-to-peer band copies) once after 2 "
                    "frames; wall time incl. the gather, after one warm-up frame"}


def load_schedules():
    """profiles/schedules.json: the launch schedule (div, mult -- mrt_get_schedule) each workload settled at in a measuring run
    on this hardware (scripts/settle_schedules.py), keyed "<config>_n<gpus>".  Pinning it (mrt_set_schedule_hint) makes two runs
    schedule alike: the controller decides on the host's wall clock and needs tens of frames (C2: 60) to get there."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "schedules.json")))
    except Exception:
        return {}


def settle(st, step, fence, agree, max_frames, max_seconds, done=0):
    """Run untimed steps until the launch schedule is final for this workload (every rank's) AND three generations of frames
    (3 x the frames in flight, `done` of them rendered already) have gone through the pipeline at it -- the first generation
    starts on an empty chip, the second still inherits its convoys -- or the cap.  `agree(ready, capped)` -> (every rank is ready,
    some rank has hit its cap): with N > 1 a step holds a collective, so all ranks must leave this loop in the same iteration
    (their clocks differ).  Returns the steps run."""
    n, t0 = 0, time.perf_counter()
    while True:
        sch = st.get_schedule()
        ready, capped = agree(bool(sch["settled"]) and done + n >= 3 * sch["frames_in_flight"],
                              n >= max_frames or time.perf_counter() - t0 >= max_seconds)
        if ready or capped:
            break
        step()
        n += 1
    fence()
    return n


def other_config_rates(M, schedules, budget_s):
    """After the headline's timed region (never inside it), N = 1: BASELINE.json's other configs and the share one GPU of an
    8-GPU C5 run renders, each on a fresh context -- pinned schedule where profiles/schedules.json has one, else measured until
    settled --, pipeline filled, then `steps` timed mrt_redraw calls; samples counted and asserted.  Short legs: a few steps each."""
    legs = [   # name, scene, width, height, spp, depth, shard, warm-up frames (pinned schedule), timed steps, schedule key
        ("c1", "default", 400, 225, 16, 8, None, 600, 3000, "c1_n1"),
        ("c2", "cover", 1200, 675, 64, 50, None, 30, 150, "c2_n1"),
        ("c4", "cover-glass", 3840, 2160, 1024, 50, None, 4, 8, "c4_n1"),
        ("c5", "stress", 1920, 1080, 4096, 50, None, 4, 8, "c5_n1"),
        # (sixteen frames in flight: the pipeline's start -- sixteen launches at once -- takes two rounds of frames to even out)
        ("c5_share_1_of_8", "stress", 1920, 1080, 4096, 50, (0, 8), 32, 96, "c5_n8"),
    ]
    out, t_begin = {}, time.perf_counter()
    for name, scene, w, h, spp, depth, shard, warm, steps, key in legs:
        if time.perf_counter() - t_begin > budget_s:
            out[name] = {"skipped": f"the {budget_s:.0f} s budget of the other configs was spent"}
            continue
        spheres, cam = (M.scene_cover(1, scene == "cover-glass") if scene.startswith("cover") else
                        M.scene_stress(1, 100) if scene == "stress" else (M.scene_default(), None))
        try:
            with M.State(M.Args(w, h, spp, depth, 1.0), seed=1, shard=shard) as s2:
                s2.set_world(spheres)
                if cam is not None:
                    s2.set_camera(cam)
                s2.set_draw_counting(False)
                pin = schedules.get(key)
                settle_frames = 0
                if pin:
                    s2.set_schedule_hint(int(pin["div"]), int(pin["mult"]))
                else:
                    settle_frames = settle(s2, s2.redraw, s2.sync, lambda ready, capped: (ready, capped), 400, 12.0)
                for _ in range(warm if pin else 0):     # fill the pipeline
                    s2.redraw()
                s2.sync()
                c0 = s2.read_counters()
                t0 = time.perf_counter()
                for _ in range(steps):
                    s2.redraw()
                s2.sync()
                dt = time.perf_counter() - t0
                c1 = s2.read_counters()
                sch = s2.get_schedule()
            n_px = w * h if shard is None else sum(min(8, h - 8 * b) * w for b in range((h + 7) // 8) if b % shard[1] == shard[0])
            assert c1["samples"] - c0["samples"] == n_px * spp * steps, (name, c1["samples"] - c0["samples"], n_px * spp * steps)
            out[name] = {"value": n_px * spp * steps / dt * 1e-6, "unit": "Msamples/s" + (" per GPU" if shard else ""),
                         "ms_per_step": dt / steps * 1e3, "steps": steps, "settled": [sch["div"], sch["mult"]],
                         "schedule_final": sch["settled"], "frames_in_flight": sch["frames_in_flight"],
                         "schedule_source": "profiles/schedules.json (pinned)" if pin else f"measured: {settle_frames} untimed frames",
                         "lane_utilisation": (c1["world_hit_calls"] - c0["world_hit_calls"]) / max(1, c1["lane_slots"] - c0["lane_slots"]),
                         "workload": f"{scene} {w}x{h}x{spp} depth {depth}" + (f", rank {shard[0]} of {shard[1]}'s interleaved 8-row bands" if shard else "")}
        except Exception as e:           # noqa: BLE001 -- a failed side leg must not take the headline line with it
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    out["seconds"] = time.perf_counter() - t_begin
    out["note"] = ("side legs after the headline's timed region, one fresh context each, one mrt_redraw per step, stream RNG; SHORT "
                   "runs from an empty pipeline to an empty pipeline (a few generations of frames: the fill and the drain cost the "
                   "long-frame configs several per cent; the full runs are profiles/r05_bench_c*.json); `value` above stays the headline's")
    return out


def tests_per_launch(hits, steps, world, n_spheres):
    """What the reference's linear scan (shader.wgsl:314-329: one sphere test per sphere per world_hit) executes per launch."""
    return hits / steps / world * n_spheres if steps else 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=0, help="override the workload (not a valid headline run)")
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--scene", default="cover-glass", choices=["cover-glass", "cover", "default", "stress"])
    ap.add_argument("--config", default="c3", choices=["c1", "c2", "c3", "c4", "c5", "interactive"],
                    help="c3 (default): the headline 1920x1080x512 cover scene, weak-scaled with N; c1 / c2 / c4 / c5: BASELINE's "
                         "configs[0] / [1] / [3] / [4] as stated; interactive: the reference's default Args (1 sample per frame, "
                         "lib.rs:27-37) on the 1920x1080 cover scene")
    ap.add_argument("--frames-per-step", type=int, default=1,
                    help="frames one step renders through mrt_render (which may share a launch among the frames of a small or short "
                         "frame); 1 (default) = one mrt_redraw per step")
    ap.add_argument("--rng", default="stream", choices=["stream", "counter"],
                    help="stream (default, every config): the reference's one Xoshiro128+ stream per pixel per frame; counter: per-sample "
                         "hashed states, a pixel's samples summed in blocks of 64 that different lanes may render (north_star's "
                         "counter-based RNG: an EXTENSION with different images, never a headline line; it keeps the pixel-starved 1/8 "
                         "shares of an 8-GPU C5 run busy, DESIGN.md 4 / 7)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--schedule", default="hint", choices=["hint", "measure"],
                    help="hint (default): pin the launch schedule profiles/schedules.json holds for this workload, if any "
                         "(mrt_set_schedule_hint); measure: let the library's controller find it -- either way the run warms up until "
                         "the schedule is final, and the line says which and what it was")
    ap.add_argument("--hint", default="", help="div,mult: pin THIS launch schedule instead (e.g. 1,1 under rocprofv3 --pmc, which runs the "
                                               "launches one at a time: a full-width launch then has the chip the counters are divided by)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1 headline run: skip the side legs after the timed region that put C1, C2, C4, C5 and C5's 1/8 share into "
                         "the line (other_configs)")
    ap.add_argument("--no-abi-legs", action="store_true",
                    help="N > 1: skip the two non-fatal legs after the timed run that drive the C ABI's own gathers (mrt_gather_rccl on a "
                         "communicator made here; native_runner --gpus N in one process), reported as abi_rccl_gather / abi_single_process")
    ap.add_argument("--verify", action="store_true", help="N > 1: fail if the verification below fails (default: only report it)")
    ap.add_argument("--no-verify", action="store_true",
                    help="N > 1: skip the check after the timed run in which the ranks render and gather one more frame and rank 0 "
                         "compares it bit for bit with the frame it renders alone (reported as gathered_image_equals_unsharded_frame)")
    a = ap.parse_args()
    force_dist = os.environ.get("MRT_BENCH_FORCE_DIST") == "1"      # rehearses the RCCL path on one GPU
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or force_dist):
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))

    # The contract is ONE JSON line on stdout.  Libraries chat there too (RCCL greets a new communicator with a version banner,
    # gloo announces its peers), so from here on file descriptor 1 points at stderr and the line goes to a private duplicate of
    # the real stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # (before anything initialises HIP: up to 8 frames of a pixel-starved shard run side by side, each needs a hardware queue)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
    import numpy as np
    import torch
    import torch.distributed as dist
    import myraytracer_amd as M
    from myraytracer_amd import dist as mdist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit(f"bench.py --gpus {a.gpus} must be launched with torch.distributed.run --nproc-per-node {a.gpus}")
        a.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU fallback for the product path")
    # MRT_BENCH_BACKEND=gloo rehearses the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, the gather goes
    # through host memory): it exercises launch, sharding, gather and --verify end to end; its timing means nothing
    backend = os.environ.get("MRT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = world > 1 or force_dist
    if use_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    width, height, spp = WORKLOADS.get(a.gpus, WORKLOADS[1])
    scaling = "weak"
    if a.config in FIXED_CONFIGS:
        a.scene, width, height, spp, a.depth = FIXED_CONFIGS[a.config]
        scaling = "strong"
    headline = (not (a.width or a.height or a.spp) and a.frames_per_step == 1 and a.rng == "stream" and
                ((a.config == "c3" and a.scene == "cover-glass" and a.depth == 50 and a.gpus in WORKLOADS) or a.config in FIXED_CONFIGS))
    width, height, spp = a.width or width, a.height or height, a.spp or spp
    seed = 1
    if a.scene == "cover-glass":
        spheres, cam = M.scene_cover(1, True)
    elif a.scene == "cover":
        spheres, cam = M.scene_cover(1, False)
    elif a.scene == "stress":
        spheres, cam = M.scene_stress(1, 100)
    else:
        spheres, cam = M.scene_default(), None

    # The State's passes and torch's ops (the RCCL gather, the un-permute) must be ordered on ONE stream.  torch's default
    # stream has the handle 0, which mrt_set_stream reads as "use your own stream" -- the gather would then not wait for the
    # frame's blend -- so the bench makes a real stream current and hands that to the State.
    stream = torch.cuda.Stream(device)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    st = M.State(M.Args(width, height, spp, a.depth, 1.0), seed=seed, device=dev_index,
                 shard=(rank, world) if world > 1 else None, stream=stream.cuda_stream)
    st.set_world(spheres)
    if cam is not None:
        st.set_camera(cam)
    if a.rng == "counter":
        st.set_rng_mode(1)
    schedules = load_schedules()
    sched_key = f"{a.config}_n{world}"
    pinned = schedules.get(sched_key) if (a.schedule == "hint" and headline and a.frames_per_step == 1) else None
    if a.hint:
        pinned = dict(zip(("div", "mult"), (int(x) for x in a.hint.split(","))))
        sched_key = "--hint " + a.hint
    if pinned:
        st.set_schedule_hint(int(pinned["div"]), int(pinned["mult"]))
    sweep_variant = st.debug_sweep_variant() or 1
    _, _, lrows, _ = st.shard_info()
    gather_device = device if backend == "nccl" else torch.device("cpu")
    staging = torch.empty((world, lrows, width, 4), dtype=torch.float32, device=gather_device) if (use_dist and rank == 0) else None

    def step():
        if a.frames_per_step == 1:
            st.redraw()                                # async on torch's current stream
        else:
            st.render(a.frames_per_step)
        if use_dist:
            local = mdist.framebuffer_tensor(st, device)
            if backend != "nccl":
                torch.cuda.synchronize(device)
                local = local.cpu()
            return mdist.gather_framebuffer(local, height, 0, staging)
        return None

    def fence():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    # Frame 0 of a fresh context (shuffle [0;4], lib.rs:422) is the frame whose whole-frame ORACLE counters are committed
    # in tests/golden/fullsize_rows.json (C3): when this run renders that workload, its first frame must reproduce them
    first_frame_check = None
    golden = None
    if world == 1 and headline and a.config == "c3" and a.warmup > 0 and a.frames_per_step == 1:      # (the checked frame is the first warm-up step)
        try:
            golden = json.load(open(os.path.join(ROOT, "tests", "golden", "fullsize_rows.json")))["configs"]["c3"]["frame0_counters"]
        except Exception:
            golden = None
    done_first = False
    if golden is not None:
        step()
        fence()
        cf = st.read_counters()
        first_frame_check = all(cf[k] == golden[k] for k in ("samples", "world_hit_calls", "rng_draws"))
        assert first_frame_check, (cf, golden)
        done_first = True
    # The timed steps run the instantiation WITHOUT the per-lane RNG draw counter (mrt_set_draw_counting: a statistic, a few VALU
    # instructions per trip of the rejection loop); samples, world_hit calls, lane slots and member tests stay counted (wave
    # totals on the scalar side) and are asserted below.  The first warm-up frame above ran WITH the draw counter.
    st.set_draw_counting(False)
    for _ in range(a.warmup - (1 if done_first else 0)):
        step()
    fence()
    # The launch schedule must be final before the clock starts: the library's controller measures its way to it over the first
    # frames (a trial = a sync of everything in flight and a different launch width), and a decision inside the timed steps would
    # make the line depend on it.  Untimed steps beyond --warmup, until every rank's schedule is final (or a cap), reported as
    # `settle_frames`; a pinned schedule is final from the first frame.  With N > 1 all ranks then take rank 0's.
    def agree(ready, capped):
        if not use_dist:
            return ready, capped
        flags = torch.tensor([1.0 if ready else 0.0, 0.0 if capped else 1.0], dtype=torch.float64, device=gather_device)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        return bool(flags[0].item() > 0.5), bool(flags[1].item() < 0.5)
    settle_frames = 0
    if a.frames_per_step == 1 and a.steps > 0:
        settle_frames = settle(st, step, fence, agree, 600, 30.0, done=a.warmup)
        if use_dist:
            before = st.get_schedule()
            shared = mdist.share_schedule(st, 0)
            if shared is not None and shared != (before["div"], before["mult"]):       # a new setting on this rank: fill its pipeline
                n_fill = st.get_schedule()["frames_in_flight"]
            else:
                n_fill = 0
            fill = torch.tensor([float(n_fill)], dtype=torch.float64, device=gather_device)
            dist.all_reduce(fill, op=dist.ReduceOp.MAX)            # (every rank steps together: a step holds a collective)
            for _ in range(int(fill.item())):
                step()
            settle_frames += int(fill.item())
            fence()
    schedule_before = st.get_schedule()
    c0 = st.read_counters()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    c1 = st.read_counters()
    schedule_after = st.get_schedule()
    kernel_ms = st.kernel_ms_history(min(a.steps, 64))
    # multi-GPU numbers with and without the gather -- a second, separately timed run of the same
    # K steps that only renders (reported as render_only_*, never as `value`)
    elapsed_render_only = None
    if use_dist:
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            st.redraw()
        fence()
        elapsed_render_only = time.perf_counter() - t0

    # whole-job numbers: max time over ranks, summed counters
    stats = torch.tensor([elapsed, sum(kernel_ms) / max(1, len(kernel_ms)), elapsed_render_only or 0.0], dtype=torch.float64, device=gather_device)
    sums = torch.tensor([c1["world_hit_calls"] - c0["world_hit_calls"], c1["samples"] - c0["samples"],
                         c1["lane_slots"] - c0["lane_slots"], c1["member_tests"] - c0["member_tests"]],
                        dtype=torch.float64, device=gather_device)
    if use_dist:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    elapsed_max, kernel_ms_max, render_only_max = float(stats[0]), float(stats[1]), float(stats[2])
    hits, samples_counted, lane_slots, member_tests = (float(x) for x in sums)

    # Verification (N > 1, after all timing): every rank restarts its accumulation and renders ONE more frame, the shards are
    # gathered as in the timed steps, and rank 0 compares the assembled image bit for bit with the frame it renders alone.
    gather_verified = None
    ref_np = None                       # rank 0: the frame one context renders alone (frame 0 of a fresh accumulation)
    if use_dist and not a.no_verify:
        st.reset()
        fence()
        one = step()
        fence()
        if rank == 0:
            with M.State(M.Args(width, height, spp, a.depth, 1.0), seed=seed, device=dev_index) as solo:
                solo.set_world(spheres)
                if cam is not None:
                    solo.set_camera(cam)
                if a.rng == "counter":
                    solo.set_rng_mode(1)
                solo.redraw()
                solo.sync()
                ref_np = solo.read_framebuffer()
                ref = torch.from_numpy(ref_np)
            gather_verified = bool(torch.equal(one.cpu().view(torch.int32), ref.view(torch.int32)))
            if a.verify:
                assert gather_verified, "the gathered image differs from the unsharded frame"
        fence()

    # which ranks RCCL actually connected: every rank reports its device; the root checks the communicator's size
    rank_devices = [None] * world
    if use_dist:
        # ... with its own share of the timed run, so that an imbalance is diagnosable from this one line
        my_hits, my_slots = c1["world_hit_calls"] - c0["world_hit_calls"], c1["lane_slots"] - c0["lane_slots"]
        dist.all_gather_object(rank_devices, {"rank": rank, "local_rank": local_rank, "device": torch.cuda.get_device_name(device),
                                              "pci_bus_id": getattr(torch.cuda.get_device_properties(device), "pci_bus_id", None),
                                              "ms_per_step": elapsed / max(1, a.steps) * 1e3,
                                              "render_only_ms_per_step": (elapsed_render_only or 0.0) / max(1, a.steps) * 1e3,
                                              "kernel_ms": sum(kernel_ms) / max(1, len(kernel_ms)),
                                              "lane_utilisation": my_hits / my_slots if my_slots else None,
                                              "world_hit_calls": my_hits})
    else:
        rank_devices = [{"rank": 0, "local_rank": local_rank, "device": torch.cuda.get_device_name(device)}]

    out = None
    if rank == 0:
        total_samples = float(width) * height * spp * a.steps * a.frames_per_step
        assert samples_counted == total_samples or a.steps == 0, (samples_counted, total_samples)
        value = total_samples / elapsed_max * 1e-6
        n_spheres = len(spheres)
        # roofline of the dominant (only) kernel, per launch on one GPU
        local_px = float(width) * height / world
        # render_kernel per launch: 16 B seed in + 16 B colour sum out per pixel, scene once.  (The blend
        # with the previous framebuffer -- 48 B per pixel -- is finalize_kernel, an HBM-bound ~25 us pass.)
        alg_bytes = local_px * (16 + 16) + ((n_spheres + 7) // 8 * 8) * 16 + n_spheres * 28
        achieved = alg_bytes / (kernel_ms_max * 1e-3) * 1e-9
        key = f"{a.scene}_{width}x{height}x{spp}_n{world}"
        traffic, pmc = None, None
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(key)
        except Exception:
            pass
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "valu_pmc.json"))).get(key)
        except Exception:
            pass
        # consecutive frames' render kernels overlap (frames in flight), so a kernel's own duration
        # (roofline.kernel_ms) exceeds the wall time per step; the VALU fractions use the wall time
        kernel_s = elapsed_max / max(1, a.steps)
        src_now = source_sha16()
        from myraytracer_amd import _lib as mlib
        lib_build_id = mlib.load().mrt_build_id().decode()
        lib_override = os.environ.get("MRT_LIB_OVERRIDE")
        # a headline needs the product binary built from the sources on disk: not a substituted library, not a stale build
        binary_ok = lib_build_id == src_now and not lib_override
        if headline and not binary_ok:
            sys.exit(f"bench.py: refusing a headline line from this binary: lib_build_id {lib_build_id} (of {mlib.LIB_PATH}) vs sources "
                     f"{src_now}, MRT_LIB_OVERRIDE={lib_override!r}; rebuild with `make`, or pass --width/--height/--spp for a custom run")
        label = "custom" if (a.width or a.height or a.spp) else a.config.upper()
        if a.rng == "counter":
            label += " (counter-RNG extension)"
        if a.frames_per_step != 1:
            label += f" ({a.frames_per_step} frames per step through mrt_render)"
        # a pixel is ONE sequential chain in the reference's RNG semantics: a rank with fewer than ~2 pixels per lane the chip
        # can hold (256 CUs x <= 20 waves x 64 lanes) is bound by its longest chains, not by throughput
        resident_lanes = 256 * 20 * 64
        if a.rng == "stream" and local_px < 2 * resident_lanes and spp >= 64:
            label += (f" [pixel-starved stream shards: {int(local_px)} pixels per rank for {resident_lanes} resident lanes, "
                      f"one sequential {spp}-sample chain per pixel]")
        prof_src = (pmc or {}).get("source_sha16")
        scene_names = {"cover-glass": "RTIOW cover scene with Dielectric + defocus blur", "cover": "RTIOW cover scene (Lambertian + Metal)",
                       "stress": "10k-sphere stress scene (100 x 100 jittered grid + ground, 80/15/5 % L/M/D)",
                       "default": "the reference's shipped 4-sphere scene"}
        out = {
            "metric": "Msamples/s (pixels x spp / s), random-spheres",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed_max / max(1, a.steps) * 1e3, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{label}: {scene_names[a.scene]} ({n_spheres} spheres, scene_seed 1), "
                                   f"{width}x{height}, {spp} spp per frame, depth {a.depth}, seed {seed}; "
                                   f"1 step = {a.frames_per_step} redraw(s) (+ RCCL gather to rank 0 when n_gpus > 1)",
                       "headline": headline, "sharding": f"interleaved 8-row bands over {world} GPU(s)",
                       "rng": {"stream": "one Xoshiro128+ stream per pixel per frame (the reference's, shader.wgsl:377-382)",
                               "counter": "per-sample hashed states, blocks of 64 samples (extension)"}[a.rng]},
            "source_sha16": src_now, "lib_build_id": lib_build_id, "lib_path": os.path.relpath(mlib.LIB_PATH, ROOT),
            "lib_matches_sources": binary_ok,
            "rccl_world_size": dist.get_world_size() if use_dist else 1, "backend": backend if use_dist else None,
            "ranks": rank_devices,
            "scene_upload_ms": st.last_set_world_ms(),
            # what the timed steps were scheduled with (mrt_get_schedule: a frame's kernel on 1 / div of the persistent waves,
            # max(2, div) x mult frames in flight)
            "schedule": {"div": schedule_after["div"], "mult": schedule_after["mult"], "final": schedule_after["settled"],
                         "frames_in_flight": schedule_after["frames_in_flight"],
                         "source": ((f"pinned: {sched_key}" if a.hint else f"pinned: profiles/schedules.json[{sched_key}]") if pinned else
                                    "measured by the library's controller during the untimed steps"),
                         "settle_frames": settle_frames,
                         "changed_during_timed_steps": (schedule_before["div"], schedule_before["mult"]) != (schedule_after["div"], schedule_after["mult"]),
                         "max_concurrent_frames": schedule_after["max_concurrent_frames"] or None},
            # north_star's roofline: HBM.  `achieved` / `frac` = algorithmic bytes of one launch over the WALL time per step (two
            # frames are in flight, so a launch's own duration -- kernel_ms, HIP events on its stream -- spans about two steps;
            # the per-launch figures are kept beside it).  Tiny by construction: 0.06 byte per sample; see valu_issue.
            "roofline": {"bound": "hbm", "achieved": alg_bytes / kernel_s * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg_bytes / kernel_s * 1e-9 / HBM_PEAK_GBS, "traffic": traffic,
                         "achieved_per_overlapped_launch": achieved, "frac_per_overlapped_launch": achieved / HBM_PEAK_GBS,
                         "traffic_source": (f"profiles/hbm_traffic.json: rocprofv3 --pmc passes of this command (profiles/README.md), NOT measured "
                                            f"in this run; profiled source {prof_src or '?'}, this run's source {src_now}") if traffic is not None else None,
                         "kernel": "mrt::render_kernel", "kernel_ms": kernel_ms_max,
                         "note": "achieved = algorithmic bytes per launch / wall time per step; kernel_ms = mean launch duration (HIP events on "
                                 "its stream): launches of consecutive frames overlap (profiles/*_launch_timeline.txt), so it is about twice "
                                 "ms_per_step and frac_per_overlapped_launch about half of frac",
                         "algorithmic_bytes_per_launch": alg_bytes},
            # the bound that binds: VALU issue slots.  frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x cycles), per serialised launch
            # under rocprofv3 --pmc (committed profile of this command); ceiling = what a pure sweep / a brute-force frame reach
            "valu_issue": ({"frac": pmc.get("issue_frac"), "peak": 1.0, "measured_ceiling": [0.865, 0.886],
                            # the same instruction count (deterministic for the workload) over THIS run's wall time per step: the
                            # chip-wide figure when launches overlap or run narrower than the chip (frac is per serialised launch)
                            "frac_over_wall_time": (pmc["valu_insts_per_launch"] * 2.0 / (1024.0 * kernel_s * SCLK_HZ)
                                                    if pmc.get("valu_insts_per_launch") else None),
                            "sclk_hz_assumed": SCLK_HZ,
                            "ceiling_source": "profiles/r01_ubench_pmc.txt (sweep microbenchmark; brute-force kernel over a frame)",
                            "thread_utilisation": pmc.get("thread_utilisation"),
                            "valu_insts_per_wave_bounce": pmc.get("valu_insts_per_wave_bounce"),
                            "source": pmc.get("source"), "profiled_source_sha16": prof_src, "this_run_source_sha16": src_now,
                            "profile_matches_this_source": prof_src == src_now,
                            "note": "NOT measured in this run: SQ counters of rocprofv3 --pmc passes of this command"} if pmc else None),
            "valu": {"note": "the binding resource is VALU issue, not HBM.  `executed_*` is what the kernel ran per launch (bound tests: "
                             "the swept top level of the bounding-sphere hierarchy; member discriminants only under candidate "
                             "bounds); `pmc_*` are SQ counters of a rocprofv3 --pmc pass of this command (profiles/valu_pmc.json, "
                             "NOT measured in this run): issue_frac = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x kernel cycles), "
                             "thread_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)",
                     "reference_sphere_tests_per_launch": tests_per_launch(hits, a.steps, world, n_spheres),
                     "sweep_variant": {1: "SGPR-fed VALU sweep", 2: "bf16-split GEMMs on the matrix cores (v_mfma_f32_32x32x16_bf16)"}[sweep_variant],
                     # matrix-core work of the sweep: per wave and world_hit (sweep_records / 32) tiles x 2 ray halves x 2
                     # GEMMs of v_mfma_f32_32x32x16_bf16 (32 x 32 x 16 x 2 flop each); dense bf16 peak 2,500 TFLOP/s
                     "mfma_bf16_tflops": (lane_slots / 64.0 * (c1["sweep_records"] / 32.0) * 4.0 * 32768.0 / max(1, a.steps) / world
                                          / kernel_s * 1e-12) if (a.steps and sweep_variant == 2) else 0.0,
                     "mfma_bf16_peak_tflops": 2500.0,
                     "mean_bounces_per_sample": hits / total_samples if total_samples else None,
                     "lane_utilisation": hits / lane_slots if lane_slots else None,
                     "executed_bound_tests_per_launch": hits / a.steps / world * c1["sweep_records"] if a.steps else 0.0,
                     "executed_member_discriminants_per_launch": member_tests / a.steps / world if a.steps else 0.0,
                     "executed_valu_issue_frac_of_sweep_and_members":
                         ((hits * c1["sweep_records"] * VALU_PER_BOUND_TEST[sweep_variant] + member_tests * VALU_PER_MEMBER_TEST)
                          / max(1, a.steps) / world / kernel_s / LANE_OPS_PEAK) if a.steps else None,
                     "pmc_issue_frac": pmc.get("issue_frac") if pmc else None,
                     "pmc_thread_utilisation": pmc.get("thread_utilisation") if pmc else None,
                     "pmc_valu_insts_per_wave_bounce": pmc.get("valu_insts_per_wave_bounce") if pmc else None},
        }
        out["statistics"] = ("timed steps: launches without the per-lane RNG draw counter (mrt_set_draw_counting(0)); samples and world_hit "
                             "calls counted and checked" + ("; the first warm-up frame ran with the draw counter and reproduced the oracle's "
                                                            "whole-frame samples / world_hit_calls / rng_draws" if first_frame_check else ""))
        if first_frame_check is not None:
            out["first_frame_counters_equal_oracle"] = first_frame_check
        if gather_verified is not None:
            out["gathered_image_equals_unsharded_frame"] = gather_verified
        if use_dist and backend != "nccl":
            out["rehearsal"] = f"backend {backend}: ranks share GPUs and gather through host memory -- a functional rehearsal, not a measurement"
        if use_dist and render_only_max > 0:
            out["render_only_value"] = total_samples / render_only_max * 1e-6
            out["render_only_ms_per_step"] = render_only_max / max(1, a.steps) * 1e3
            out["render_only_note"] = "the same K steps without the gather to rank 0, timed separately after the headline run"
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spheres, cam, width, height, a.depth, seed)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        if world == 1 and not use_dist and headline and a.config == "c3" and not a.no_other_configs and a.steps > 0:
            st.close()          # (everything of the headline's context has been read; its streams need not sit beside the side legs')
            out["other_configs"] = other_config_rates(M, schedules if a.schedule == "hint" else {}, 90.0)

    # ---- N > 1: the C ABI's own gathers on the same workload, after everything that is timed for `value`.  Non-fatal by
    # construction: any exception becomes {"error": ...}.  A HANG (a collective or a peer copy that never completes -- these
    # legs have never run between two physical GPUs) is ended by a watchdog on EVERY rank: rank 0 first prints the line it
    # already holds -- `value` is final at this point -- with the hang recorded in it ("hang": which call was in flight), every
    # rank names the call on stderr, and all exit with HANG_EXIT_CODE.  The same watchdog covers the final barrier /
    # destroy_process_group, where a rank that died in a leg would otherwise leave the others waiting for RCCL's own timeout.
    # Exit status: 0 whenever the JSON line carries a measured `value` and nothing hung; HANG_EXIT_CODE (75) after a hang -- the
    # line is still complete and says so -- so that a hang is never recorded as a clean run; other non-zero codes: no line.
    import threading
    printed = threading.Lock()
    in_flight = {"call": None}

    def emit(extra=None):
        if rank == 0 and out is not None and printed.acquire(blocking=False):
            if extra:
                out.update(extra)
            os.write(json_fd, (json.dumps(out) + "\n").encode())

    def watchdog(seconds, leg):
        def give_up():
            msg = f"bench.py rank {rank}: watchdog after {seconds:.0f} s in {leg}; call in flight: {in_flight['call']}"
            os.write(2, (msg + "\n").encode())
            emit({leg: {"error": f"timed out after {seconds:.0f} s (watchdog); `value` above was already final"},
                  "hang": {"leg": leg, "call_in_flight": in_flight["call"], "rank": rank, "exit_code": HANG_EXIT_CODE}})
            os._exit(HANG_EXIT_CODE)
        t = threading.Timer(seconds, give_up)
        t.daemon = True
        t.start()
        return t

    if use_dist and backend == "nccl" and not a.no_abi_legs:
        dog = watchdog(240.0, "abi_rccl_gather")
        leg = None
        try:
            leg = abi_rccl_leg(st, dist, world, rank, min(a.steps, 5), fence,
                               ref_np.view(np.uint32) if ref_np is not None else None, in_flight)
        except Exception as e:          # noqa: BLE001 -- non-fatal by design
            leg = {"error": f"{type(e).__name__}: {e}"[:300], "call_in_flight": in_flight["call"]}
        dog.cancel()
        if rank == 0:
            out["abi_rccl_gather"] = leg
    st.close()
    if use_dist:
        dog = watchdog(120.0, "teardown")
        try:
            in_flight["call"] = "dist.barrier"
            dist.barrier()
            in_flight["call"] = "dist.destroy_process_group"
            dist.destroy_process_group()
        except Exception:               # noqa: BLE001
            pass
        dog.cancel()
    if rank == 0 and use_dist and world > 1 and not a.no_abi_legs and (backend == "nccl" or os.environ.get("MRT_BENCH_ABI_DEVICES")):
        dog2 = watchdog(300.0, "abi_single_process")
        try:
            in_flight["call"] = "native_runner --gpus N (mrt_gather: hipMemcpyPeerAsync per band)"
            out["abi_single_process"] = abi_single_process_leg(a, world, width, height, spp,
                                                               ref_np[..., :3].copy().view(np.uint32) if ref_np is not None else None)
        except Exception as e:          # noqa: BLE001
            out["abi_single_process"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        dog2.cancel()
    emit()
    if rank == 0 and out is None:
        sys.exit(1)


if __name__ == "__main__":
    main()
