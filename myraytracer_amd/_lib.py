"""ctypes binding of the C ABI declared in include/myraytracer_amd.h.

The shared library is built in-tree (myraytracer_amd/lib/libmyraytracer_amd.so, see the
top-level Makefile).  There is no fallback of any kind: a missing library raises
ImportError, a missing GPU makes mrt_create fail with MRT_ERR_NO_DEVICE.
"""
import ctypes as C
import importlib.util
import os
import sys

# Up to 8 frames of a pixel-starved shard run at a time, each on a HIP stream of its own (api.cpp, redraw_frames); they only
# run side by side on hardware queues of their own, and HIP's default is 4 per process.  Read by the runtime when it
# initialises -- set here, at import, before anything (torch included) has made a HIP call.  Never overrides the caller's value.
# (The C library itself never touches the environment: it measures what it got and holds its schedule to that.)
# Twenty: enough for the library's up to 16 side streams + the context's own + the null stream to have a queue each (with 16,
# two of sixteen frames in flight take turns on one queue: C5's 1/8 share 2,790 instead of 3,750 Msamples/s) -- and not more:
# once a process has created some 24 hardware queues, EVERY kernel of it runs slower, for good -- measured, round 5
# (scripts/queue_oversubscription.py, profiles/r05_queue_oversubscription.txt): C2 13,400 Msamples/s with up to 22 queues,
# 12,000 with 24, 9,060 with 32, and closing the streams does not bring it back.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MRT_LIB_OVERRIDE") or os.path.join(_HERE, "lib", "libmyraytracer_amd.so")

# every symbol include/myraytracer_amd.h (the boundary) and include/myraytracer_amd_debug.h (diagnostics, mrt_debug_*)
# declare (tests/test_abi.py checks this list against both headers and against the loaded library)
EXPORTS = [
    "mrt_args_default", "mrt_args_resolve_size", "mrt_create", "mrt_destroy", "mrt_set_shard",
    "mrt_set_stream", "mrt_set_world_raw", "mrt_set_world", "mrt_pack_world", "mrt_set_camera",
    "mrt_camera_derive", "mrt_set_seeds", "mrt_read_seeds", "mrt_redraw", "mrt_render", "mrt_sync",
    "mrt_reset", "mrt_get_locals", "mrt_set_rng_shuffle", "mrt_set_samples_per_frame", "mrt_set_rng_mode",
    "mrt_frames_done", "mrt_frame_weight", "mrt_frame_shuffle", "mrt_pixel_seed", "mrt_shard_info",
    "mrt_framebuffer_device_ptr", "mrt_read_framebuffer", "mrt_read_counters", "mrt_last_kernel_ms",
    "mrt_kernel_ms_history", "mrt_debug_read_counters", "mrt_debug_wave_log", "mrt_debug_set_tile_sort", "mrt_debug_set_cluster_factor", "mrt_debug_set_hierarchy", "mrt_debug_set_sweep", "mrt_debug_sweep_variant", "mrt_debug_build_hierarchy", "mrt_debug_read_pixel_costs", "mrt_debug_set_schedule", "mrt_debug_mfma_scale",
    "mrt_last_error", "mrt_status_string", "mrt_abi_version", "mrt_build_id", "mrt_scene_default", "mrt_scene_cover",
    "mrt_scene_stress", "mrt_scene_save", "mrt_scene_load", "mrt_write_pfm", "mrt_write_ppm",
    "mrt_srgb8", "mrt_write_png", "mrt_gather", "mrt_gather_rccl", "mrt_gathered_device_ptr", "mrt_read_gathered",
    "mrt_shard_global_row", "mrt_shard_local_rows", "mrt_unshard_rows", "mrt_debug_last_set_world_ms", "mrt_debug_world_hit", "mrt_debug_set_frame_batching", "mrt_debug_set_gather_per_band", "mrt_debug_arith", "mrt_debug_arith_pairs", "mrt_debug_set_boxes", "mrt_debug_build_boxes", "mrt_set_draw_counting", "mrt_debug_last_launch", "mrt_debug_set_frames_in_flight", "mrt_debug_lds_layout", "mrt_debug_build_boxes_top_down",
    "mrt_set_wait_timeout", "mrt_get_schedule", "mrt_set_schedule_hint", "mrt_debug_width_policy", "mrt_debug_stream_concurrency", "mrt_debug_wave_log_frame",
]


class MrtArgs(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("samples_per_frame", C.c_uint32),
                ("ray_depth", C.c_uint32), ("max_framebuffer_weight", C.c_float)]


class MrtLocals(C.Structure):
    _fields_ = [("shape", C.c_uint32 * 2), ("samples_per_frame", C.c_uint32), ("ray_depth", C.c_uint32),
                ("rng_shuffle", C.c_uint32 * 4), ("framebuffer_weight", C.c_float), ("rng_mode", C.c_uint32),
                ("_padding", C.c_uint32 * 2)]


class MrtSphereRange(C.Structure):
    _fields_ = [("center_base_idx", C.c_int32), ("radius_base_idx", C.c_int32),
                ("material_ty_base_idx", C.c_int32), ("material_idx_base_idx", C.c_int32),
                ("length", C.c_int32), ("_padding", C.c_int32 * 3)]


class MrtLambertianRange(C.Structure):
    _fields_ = [("albedo_base_idx", C.c_int32), ("length", C.c_int32), ("_padding", C.c_int32 * 2)]


class MrtMetalRange(C.Structure):
    _fields_ = [("albedo_base_idx", C.c_int32), ("fuzz_base_idx", C.c_int32), ("length", C.c_int32),
                ("_padding", C.c_int32)]


class MrtDielectricRange(C.Structure):
    _fields_ = [("ior_base_idx", C.c_int32), ("length", C.c_int32), ("_padding", C.c_int32 * 2)]


class MrtWorld(C.Structure):
    _fields_ = [("spheres", MrtSphereRange), ("lambertians", MrtLambertianRange),
                ("metals", MrtMetalRange), ("dielectrics", MrtDielectricRange)]


class MrtSphere(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("material_ty", C.c_int32),
                ("albedo", C.c_float * 3), ("param", C.c_float)]


class MrtCamera(C.Structure):
    _fields_ = [("mode", C.c_int32), ("lookfrom", C.c_float * 3), ("lookat", C.c_float * 3),
                ("vup", C.c_float * 3), ("vfov_deg", C.c_float), ("defocus_angle_deg", C.c_float),
                ("focus_dist", C.c_float)]


class MrtCameraRaw(C.Structure):
    _fields_ = [("mode", C.c_int32), ("defocus", C.c_int32), ("origin", C.c_float * 3), ("su", C.c_float * 3),
                ("sv", C.c_float * 3), ("fw", C.c_float * 3), ("ru", C.c_float * 3), ("rv", C.c_float * 3)]


class MrtCounters(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("world_hit_calls", C.c_uint64), ("rng_draws", C.c_uint64),
                ("lane_slots", C.c_uint64), ("member_tests", C.c_uint64), ("sweep_records", C.c_uint64)]


_lib = None


def _soname(path):
    """DT_SONAME of a 64-bit little-endian ELF shared object (None if it cannot be read)."""
    import struct
    try:
        with open(path, "rb") as f:
            head = f.read(64)
            if head[:5] != b"\x7fELF\x02" or head[5] != 1:
                return None
            shoff, = struct.unpack_from("<Q", head, 0x28)
            shentsize, shnum = struct.unpack_from("<HH", head, 0x3A)
            f.seek(shoff)
            raw = f.read(shentsize * shnum)
            secs = [struct.unpack_from("<IIQQQQIIQQ", raw, i * shentsize) for i in range(shnum)]
            for sec in secs:
                if sec[1] != 6:                             # SHT_DYNAMIC
                    continue
                strtab = secs[sec[6]]                       # sh_link -> .dynstr
                f.seek(sec[4])
                dyn = f.read(sec[5])
                for off in range(0, len(dyn) - 15, 16):
                    tag, val = struct.unpack_from("<qQ", dyn, off)
                    if tag == 14:                           # DT_SONAME
                        f.seek(strtab[4] + val)
                        name = f.read(256)
                        return name[:name.index(b"\0")].decode()
                    if tag == 0:
                        break
    except (OSError, struct.error, ValueError, IndexError):
        pass
    return None


def _needed_hip_soname():
    """The libamdhip64 soname our library was linked against (DT_NEEDED), e.g. libamdhip64.so.7."""
    import re
    try:
        m = re.search(rb"libamdhip64\.so\.\d+", open(LIB_PATH, "rb").read())
        return m.group(0).decode() if m else None
    except OSError:
        return None


def _one_hip_runtime():
    """A process must run ONE HIP / HSA runtime.  PyTorch-ROCm bundles its own (torch/lib/libamdhip64.so, the same soname as
    /opt/rocm's): whichever copy is loaded first serves both torch and this library, and when /opt/rocm's comes first torch's
    other bundled libraries still bring their own HSA runtime along -- torch then reports "No HIP GPUs are available".  So if
    torch is installed but not imported yet, its runtime is loaded here, before ours resolves its libamdhip64 -- but only if the
    bundled library's DT_SONAME is the one ours needs (a torch wheel of another ROCm major would otherwise put two runtimes,
    or interposed symbols of the wrong one, into the process: the very failure this is meant to prevent); a skipped preload is
    reported on stderr when MRT_VERBOSE is set.  MRT_HIP_RUNTIME=system skips this (a process without torch)."""
    if os.environ.get("MRT_HIP_RUNTIME") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    hip = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(hip):
        return
    have, need = _soname(os.path.realpath(hip)), _needed_hip_soname()
    why = None
    if have is None or need is None or have != need:
        why = f"its soname {have!r} is not the {need!r} {LIB_PATH} was linked against"
    else:
        try:
            C.CDLL(hip, mode=C.RTLD_GLOBAL)
        except OSError as e:
            why = str(e)
    if why and os.environ.get("MRT_VERBOSE"):
        print(f"myraytracer_amd: torch's bundled {hip} not preloaded: {why}", file=sys.stderr)


def load():
    """Load the in-tree shared library (once) and declare the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    _one_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make` at the repo root "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "myraytracer_amd has no Python or CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, f32, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_float, C.c_size_t
    P = C.POINTER
    sig = {
        "mrt_args_default": (None, [P(MrtArgs)]),
        "mrt_args_resolve_size": (None, [P(MrtArgs)]),
        "mrt_create": (i32, [P(MrtArgs), u64, i32, P(vp)]),
        "mrt_destroy": (None, [vp]),
        "mrt_set_shard": (i32, [vp, u32, u32]),
        "mrt_set_stream": (i32, [vp, vp]),
        "mrt_set_world_raw": (i32, [vp, vp, sz, vp, sz, vp, sz, vp, sz]),
        "mrt_set_world": (i32, [vp, vp, sz]),
        "mrt_pack_world": (i32, [vp, sz, P(MrtWorld), vp, sz, P(sz), vp, sz, P(sz), vp, sz, P(sz)]),
        "mrt_set_camera": (i32, [vp, P(MrtCamera)]),
        "mrt_camera_derive": (i32, [P(MrtCamera), P(MrtCameraRaw)]),
        "mrt_set_seeds": (i32, [vp, vp, sz]),
        "mrt_read_seeds": (i32, [vp, vp, sz]),
        "mrt_redraw": (i32, [vp]),
        "mrt_render": (i32, [vp, u32]),
        "mrt_sync": (i32, [vp]),
        "mrt_reset": (i32, [vp]),
        "mrt_get_locals": (i32, [vp, P(MrtLocals)]),
        "mrt_set_rng_shuffle": (i32, [vp, P(u32)]),
        "mrt_set_samples_per_frame": (i32, [vp, u32]),
        "mrt_set_rng_mode": (i32, [vp, u32]),
        "mrt_frames_done": (u32, [vp]),
        "mrt_frame_weight": (f32, [u32, f32]),
        "mrt_frame_shuffle": (None, [u64, u32, P(u32)]),
        "mrt_pixel_seed": (None, [u64, u64, P(u32)]),
        "mrt_shard_info": (i32, [vp, P(u32), P(u32), P(u32), P(u32)]),
        "mrt_framebuffer_device_ptr": (vp, [vp]),
        "mrt_read_framebuffer": (i32, [vp, vp, sz]),
        "mrt_read_counters": (i32, [vp, P(MrtCounters)]),
        "mrt_last_kernel_ms": (i32, [vp, P(f32)]),
        "mrt_kernel_ms_history": (i32, [vp, P(f32), sz, P(sz)]),
        "mrt_debug_read_counters": (i32, [vp, P(u64)]),
        "mrt_debug_wave_log": (i32, [vp, vp, sz, P(sz)]),
        "mrt_debug_set_tile_sort": (i32, [vp, i32]),
        "mrt_debug_set_cluster_factor": (i32, [vp, f32]),
        "mrt_debug_set_hierarchy": (i32, [vp, u32, u32]),
        "mrt_debug_set_sweep": (i32, [vp, i32]),
        "mrt_debug_sweep_variant": (i32, [vp]),
        "mrt_debug_build_hierarchy": (i32, [vp, sz, u32, u32, vp, sz, vp, sz, vp, sz, vp, sz, vp, vp]),
        "mrt_debug_read_pixel_costs": (i32, [vp, vp, sz]),
        "mrt_debug_set_schedule": (i32, [vp, u32, i32]),
        "mrt_debug_mfma_scale": (i32, [C.c_double, vp, vp]),
        "mrt_last_error": (C.c_char_p, [vp]),
        "mrt_status_string": (C.c_char_p, [i32]),
        "mrt_abi_version": (i32, []),
        "mrt_build_id": (C.c_char_p, []),
        "mrt_scene_default": (i32, [vp, sz]),
        "mrt_scene_cover": (i32, [u64, i32, vp, sz, P(MrtCamera)]),
        "mrt_scene_stress": (i32, [u64, u32, vp, sz, P(MrtCamera)]),
        "mrt_scene_save": (i32, [C.c_char_p, vp, sz, P(MrtCamera)]),
        "mrt_scene_load": (i32, [C.c_char_p, vp, sz, P(MrtCamera), P(C.c_int)]),
        "mrt_write_pfm": (i32, [C.c_char_p, vp, u32, u32]),
        "mrt_write_ppm": (i32, [C.c_char_p, vp, u32, u32]),
        "mrt_srgb8": (C.c_uint8, [f32]),
        "mrt_write_png": (i32, [C.c_char_p, vp, u32, u32]),
        "mrt_gather": (i32, [P(vp), u32, u32]),
        "mrt_gather_rccl": (i32, [vp, vp, u32]),
        "mrt_gathered_device_ptr": (vp, [vp]),
        "mrt_read_gathered": (i32, [vp, vp, sz]),
        "mrt_shard_global_row": (u32, [u32, u32, u32]),
        "mrt_shard_local_rows": (u32, [u32, u32]),
        "mrt_unshard_rows": (i32, [vp, u32, u32, u32, vp]),
        "mrt_debug_last_set_world_ms": (i32, [vp, P(f32)]),
        "mrt_debug_world_hit": (i32, [vp, vp, sz, vp, vp, sz]),
        "mrt_debug_set_frame_batching": (i32, [vp, i32]),
        "mrt_debug_set_gather_per_band": (i32, [vp, i32]),
        "mrt_debug_arith": (i32, [vp, i32, P(u32), u64, u64, P(u64)]),
        "mrt_debug_arith_pairs": (i32, [vp, vp, vp, sz, vp]),
        "mrt_debug_set_boxes": (i32, [vp, i32]),
        "mrt_debug_build_boxes": (i32, [vp, sz, u32, u32, vp, sz, vp]),
        "mrt_set_draw_counting": (i32, [vp, i32]),
        "mrt_debug_last_launch": (i32, [vp, P(u32)]),
        "mrt_debug_set_frames_in_flight": (i32, [vp, i32]),
        "mrt_debug_lds_layout": (i32, [u32, u32, u32, u32, P(u32)]),
        "mrt_debug_build_boxes_top_down": (i32, [vp, sz, u32, u32, i32, vp, sz, vp]),
        "mrt_set_wait_timeout": (i32, [vp, C.c_double]),
        "mrt_get_schedule": (i32, [vp, P(u32)]),
        "mrt_set_schedule_hint": (i32, [vp, u32, u32]),
        "mrt_debug_width_policy": (i32, [i32, P(u32), P(u32), C.c_double, C.c_double]),
        "mrt_debug_stream_concurrency": (i32, [vp, u32, P(f32)]),
        "mrt_debug_wave_log_frame": (i32, [vp, u32, vp, sz, P(sz)]),
    }
    assert sorted(sig) == sorted(EXPORTS)
    for name, (res, args) in sig.items():
        fn = getattr(L, name)          # AttributeError here = the .so does not export the header's symbol
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L
