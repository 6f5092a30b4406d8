#!/usr/bin/env python3
"""Benchmark of the render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one `State::redraw` (raytracer/src/lib.rs:241-307): a full raytrace pass of
samples_per_frame spp over the whole image, blended into the accumulated framebuffer, and
-- for N > 1 -- the RCCL gather of every rank's bands to rank 0.

Workload (BASELINE.json): the metric is quoted on "1920x1080 random-spheres" = configs[2]
(C3: RTIOW cover scene with Dielectric + defocus blur, 1920x1080, 512 spp, depth 50).
Multi-GPU is weak scaling towards configs[3] (C4 = 8 x C3's samples at 8 GPUs): the same scene
and camera with N x C3's samples -- N=1 1920x1080x512, N=2 2716x1528x512 (2.001 x the pixels, same
16:9 framing), N=4 3840x2160x512, N=8 3840x2160x1024 (= C4); the image is tile-sharded in
interleaved 8-row bands, every rank renders (1/N) of it.

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

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector
LANE_OPS_PEAK = 256 * 4 * 32 * 2.4e9   # CUs x SIMDs x lanes/clk x Hz: fp32 VALU lane-ops/s
FLOP_PER_TEST = 23           # algorithmic flop per ray-sphere test (unfused count)
VALU_PER_BOUND_TEST = {1: 11, 2: 2}   # sweep variant 1: 10 fp32 VALU + 1 v_alignbit from SGPRs; 2: 1 fma + 1 v_alignbit after the matrix cores
VALU_PER_MEMBER_TEST = 13    # 11 fp32 VALU + compare + queue bookkeeping per member discriminant

WORKLOADS = {   # n_gpus -> (width, height, spp)
    1: (1920, 1080, 512),
    2: (2716, 1528, 512),
    4: (3840, 2160, 512),
    8: (3840, 2160, 1024),
}


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
    """Time the CPU oracle (kind "port") on a bounded sample of the same workload."""
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
            "sample": f"{width}x{height}, {spp} spp of the same scene/camera/depth (oracle/rt_oracle.c, "
                      f"OpenMP over rows), {dt:.1f} s; rate is spp-independent, so no extrapolation is applied"}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

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
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("MRT_BENCH_FORCE_DIST") == "1"   # the env var rehearses the RCCL path on one GPU
    if use_dist:
        dist.init_process_group("nccl", device_id=device)

    width, height, spp = WORKLOADS.get(a.gpus, WORKLOADS[1])
    headline = not (a.width or a.height or a.spp) and a.scene == "cover-glass" and a.depth == 50 and a.gpus in WORKLOADS
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

    stream = torch.cuda.current_stream(device)
    st = M.State(M.Args(width, height, spp, a.depth, 1.0), seed=seed, device=local_rank,
                 shard=(rank, world) if world > 1 else None, stream=stream.cuda_stream)
    st.set_world(spheres)
    if cam is not None:
        st.set_camera(cam)
    sweep_variant = st.debug_sweep_variant() or 1
    _, _, lrows, _ = st.shard_info()
    staging = torch.empty((world, lrows, width, 4), dtype=torch.float32, device=device) if (use_dist and rank == 0) else None

    def step():
        st.redraw()                                    # async on torch's current stream
        if use_dist:
            return mdist.gather_framebuffer(mdist.framebuffer_tensor(st, device), height, 0, staging)
        return None

    def fence():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(a.warmup):
        step()
    fence()
    c0 = st.read_counters()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    c1 = st.read_counters()
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
    stats = torch.tensor([elapsed, sum(kernel_ms) / max(1, len(kernel_ms)), elapsed_render_only or 0.0], dtype=torch.float64, device=device)
    sums = torch.tensor([c1["world_hit_calls"] - c0["world_hit_calls"], c1["samples"] - c0["samples"],
                         c1["lane_slots"] - c0["lane_slots"], c1["member_tests"] - c0["member_tests"]],
                        dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    elapsed_max, kernel_ms_max, render_only_max = float(stats[0]), float(stats[1]), float(stats[2])
    hits, samples_counted, lane_slots, member_tests = (float(x) for x in sums)

    if rank == 0:
        total_samples = float(width) * height * spp * a.steps
        assert samples_counted == total_samples or a.steps == 0, (samples_counted, total_samples)
        value = total_samples / elapsed_max * 1e-6
        n_spheres = len(spheres)
        # roofline of the dominant (only) kernel, per launch on one GPU
        local_px = float(width) * height / world
        # render_kernel per launch: 16 B seed in + 16 B colour sum out per pixel, scene once.  (The blend
        # with the previous framebuffer -- 48 B per pixel -- is finalize_kernel, an HBM-bound ~25 us pass.)
        alg_bytes = local_px * (16 + 16) + ((n_spheres + 7) // 8 * 8) * 16 + n_spheres * 28
        achieved = alg_bytes / (kernel_ms_max * 1e-3) * 1e-9
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
            traffic = tj.get(f"{a.scene}_{width}x{height}x{spp}_n{world}")
        except Exception:
            pass
        tests_per_launch = hits / a.steps / world * n_spheres if a.steps else 0.0
        # consecutive frames' render kernels overlap (two frames in flight), so a kernel's own duration
        # (roofline.kernel_ms) exceeds the wall time per step; the VALU fractions use the wall time
        kernel_s = elapsed_max / max(1, a.steps)
        out = {
            "metric": "Msamples/s (pixels x spp / s), random-spheres",
            "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed_max / max(1, a.steps) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"RTIOW cover scene ({n_spheres} spheres, scene_seed 1) "
                                   f"{'with Dielectric + defocus blur' if a.scene == 'cover-glass' else a.scene}, "
                                   f"{width}x{height}, {spp} spp per frame, depth {a.depth}, seed {seed}; "
                                   f"1 step = 1 redraw (+ RCCL gather to rank 0 when n_gpus > 1)",
                       "headline": headline, "sharding": f"interleaved 8-row bands over {world} GPU(s)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "mrt::render_kernel", "kernel_ms": kernel_ms_max,
                         "note": "kernel_ms = mean launch duration (HIP events on its stream); launches of consecutive "
                                 "frames overlap, so it is longer than ms_per_step",
                         "algorithmic_bytes_per_launch": alg_bytes},
            "valu": {"note": "the binding resource is VALU issue.  `algorithmic_*` = what the reference's linear scan "
                             "(one test per sphere per world_hit, 23 flop each) would execute; the kernel sweeps the top level "
                             "of a bounding-sphere hierarchy and evaluates member discriminants only under candidate bounds, so "
                             "the algorithmic rate may exceed the executed one and the fp32 peak; `executed_*` is what the "
                             "kernel ran (bound tests: the swept top level only)",
                     "algorithmic_sphere_tests_per_launch": tests_per_launch,
                     "algorithmic_tflops": tests_per_launch * FLOP_PER_TEST / kernel_s * 1e-12, "peak_tflops": FP32_PEAK_TFLOPS,
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
                          / max(1, a.steps) / world / kernel_s / LANE_OPS_PEAK) if a.steps else None},
        }
        if use_dist and render_only_max > 0:
            out["render_only_value"] = total_samples / render_only_max * 1e-6
            out["render_only_ms_per_step"] = render_only_max / max(1, a.steps) * 1e3
            out["render_only_note"] = "the same K steps without the gather to rank 0, timed separately after the headline run"
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(spheres, cam, width, height, a.depth, seed)
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    st.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
