"""Host-side mirror of the reference's interface for the render path, over the C ABI.

Names follow raytracer/src/lib.rs: `Args` (lib.rs:18-37), the scene description
`Lambertian` / `Metal` / `Sphere` / `World` (the fn-local `api` module, lib.rs:611-639;
`Dielectric` is the extension), and `State` with `redraw()` (lib.rs:206-308).  The
reference's toolchain (Rust) is not in this image, so this mirror is Python over ctypes;
INTEGRATION.md shows the Rust binding of the same ABI.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import MrtArgs, MrtCamera, MrtCameraRaw, MrtCounters, MrtLocals, MrtSphere, MrtWorld

LAMBERTIAN, METAL, DIELECTRIC = 1, 2, 3          # raw::MaterialTy, lib.rs:644-648 (+ extension)
SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("material_ty", "<i4"),
                         ("albedo", "<f4", 3), ("param", "<f4")])
assert SPHERE_DTYPE.itemsize == C.sizeof(MrtSphere) == 36


class MrtError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        name = _lib.load().mrt_status_string(status).decode()
        super().__init__(f"{where}: {name}" + (f" ({detail})" if detail else ""))


@dataclass
class Args:
    """raytracer::Args (lib.rs:18-37), same defaults."""
    width: int = 0
    height: int = 0
    samples_per_frame: int = 1
    ray_depth: int = 50
    max_framebuffer_weight: float = 1.0

    def resolved(self) -> "Args":
        """The size rule of App::resumed (lib.rs:113-134)."""
        a = MrtArgs(self.width, self.height, self.samples_per_frame, self.ray_depth, self.max_framebuffer_weight)
        _lib.load().mrt_args_resolve_size(C.byref(a))
        return Args(a.width, a.height, a.samples_per_frame, a.ray_depth, a.max_framebuffer_weight)

    def _c(self) -> MrtArgs:
        return MrtArgs(self.width, self.height, self.samples_per_frame, self.ray_depth, self.max_framebuffer_weight)


@dataclass
class Lambertian:          # api::Lambertian, lib.rs:613-616
    albedo: Tuple[float, float, float]


@dataclass
class Metal:               # api::Metal, lib.rs:618-622
    albedo: Tuple[float, float, float]
    fuzz: float


@dataclass
class Dielectric:          # extension (material type 3)
    ior: float


@dataclass
class Sphere:              # api::Sphere, lib.rs:630-635
    center: Tuple[float, float, float]
    radius: float
    material: object


@dataclass
class World:               # api::World, lib.rs:637-639
    spheres: List[Sphere] = field(default_factory=list)

    def to_array(self) -> np.ndarray:
        out = np.zeros(len(self.spheres), SPHERE_DTYPE)
        for i, s in enumerate(self.spheres):
            m = s.material
            if isinstance(m, Lambertian):
                out[i] = (s.center, s.radius, LAMBERTIAN, m.albedo, 0.0)
            elif isinstance(m, Metal):
                out[i] = (s.center, s.radius, METAL, m.albedo, m.fuzz)
            elif isinstance(m, Dielectric):
                out[i] = (s.center, s.radius, DIELECTRIC, (1.0, 1.0, 1.0), m.ior)
            else:
                raise TypeError(f"unknown material {m!r}")
        return out


@dataclass
class Camera:
    """mode 0 = the reference's fixed pinhole (shader.wgsl:360-381); mode 1 = look-at thin lens."""
    mode: int = 0
    lookfrom: Sequence[float] = (0.0, 0.0, 0.0)
    lookat: Sequence[float] = (0.0, 0.0, -1.0)
    vup: Sequence[float] = (0.0, 1.0, 0.0)
    vfov_deg: float = 90.0
    defocus_angle_deg: float = 0.0
    focus_dist: float = 1.0

    def _c(self) -> MrtCamera:
        c = MrtCamera()
        c.mode = self.mode
        c.lookfrom[:] = list(self.lookfrom)
        c.lookat[:] = list(self.lookat)
        c.vup[:] = list(self.vup)
        c.vfov_deg, c.defocus_angle_deg, c.focus_dist = self.vfov_deg, self.defocus_angle_deg, self.focus_dist
        return c

    @staticmethod
    def _from_c(c: MrtCamera) -> "Camera":
        return Camera(c.mode, tuple(c.lookfrom), tuple(c.lookat), tuple(c.vup), c.vfov_deg,
                      c.defocus_angle_deg, c.focus_dist)


# ------------------------------------------------------------------ host-only helpers

def pack_world(spheres: np.ndarray):
    """lib.rs:722-799 through mrt_pack_world -> (MrtWorld, vec4[n,4], f32[n], i32[n])."""
    L = _lib.load()
    spheres = np.ascontiguousarray(spheres, SPHERE_DTYPE)
    n = len(spheres)
    vec4 = np.zeros((2 * n + 1, 4), np.float32)
    f32 = np.zeros(2 * n + 1, np.float32)
    i32 = np.zeros(2 * n + 1, np.int32)
    w = MrtWorld()
    nv, nf, ni = C.c_size_t(), C.c_size_t(), C.c_size_t()
    st = L.mrt_pack_world(spheres.ctypes.data, n, C.byref(w), vec4.ctypes.data, 2 * n + 1, C.byref(nv),
                          f32.ctypes.data, 2 * n + 1, C.byref(nf), i32.ctypes.data, 2 * n + 1, C.byref(ni))
    if st:
        raise MrtError(st, "mrt_pack_world")
    return w, vec4[:nv.value].copy(), f32[:nf.value].copy(), i32[:ni.value].copy()


def camera_derive(cam: Camera) -> MrtCameraRaw:
    raw = MrtCameraRaw()
    st = _lib.load().mrt_camera_derive(C.byref(cam._c()), C.byref(raw))
    if st:
        raise MrtError(st, "mrt_camera_derive")
    return raw


def frame_weight(frames_done: int, max_w: float) -> float:
    return float(_lib.load().mrt_frame_weight(frames_done, max_w))


def frame_shuffle(seed: int, frame: int) -> List[int]:
    out = (C.c_uint32 * 4)()
    _lib.load().mrt_frame_shuffle(seed, frame, out)
    return [int(x) for x in out]


def pixel_seed(seed: int, pixel_index: int) -> List[int]:
    out = (C.c_uint32 * 4)()
    _lib.load().mrt_pixel_seed(seed, pixel_index, out)
    return [int(x) for x in out]


def scene_default() -> np.ndarray:
    """The shipped 4-sphere scene (lib.rs:687-720)."""
    out = np.zeros(4, SPHERE_DTYPE)
    n = _lib.load().mrt_scene_default(out.ctypes.data, 4)
    assert n == 4
    return out


def scene_cover(scene_seed: int = 1, dielectric: bool = False):
    out = np.zeros(512, SPHERE_DTYPE)
    cam = MrtCamera()
    n = _lib.load().mrt_scene_cover(scene_seed, int(dielectric), out.ctypes.data, len(out), C.byref(cam))
    if n < 0 or n > len(out):
        raise MrtError(-n if n < 0 else 6, "mrt_scene_cover")
    return out[:n].copy(), Camera._from_c(cam)


def scene_stress(scene_seed: int = 1, n_side: int = 100):
    out = np.zeros(n_side * n_side + 1, SPHERE_DTYPE)
    cam = MrtCamera()
    n = _lib.load().mrt_scene_stress(scene_seed, n_side, out.ctypes.data, len(out), C.byref(cam))
    if n < 0 or n > len(out):
        raise MrtError(-n if n < 0 else 6, "mrt_scene_stress")
    return out[:n].copy(), Camera._from_c(cam)


def save_scene(path: str, spheres, cam: Optional[Camera] = None):
    """Write a scene file (format: include/myraytracer_amd.h, mrt_scene_save).  cam=None writes no camera line."""
    arr = spheres.to_array() if isinstance(spheres, World) else np.ascontiguousarray(spheres, SPHERE_DTYPE)
    c = cam._c() if cam is not None else None
    st = _lib.load().mrt_scene_save(os.fsencode(path), arr.ctypes.data, len(arr), C.byref(c) if c is not None else None)
    if st != 0:
        raise MrtError(st, "mrt_scene_save", _lib.load().mrt_last_error(None).decode())


def load_scene(path: str):
    """Read a scene file -> (spheres, camera or None if the file has no camera line)."""
    L = _lib.load()
    cam, has = MrtCamera(), C.c_int(0)
    n = L.mrt_scene_load(os.fsencode(path), None, 0, C.byref(cam), C.byref(has))
    if n < 0:
        raise MrtError(-n, "mrt_scene_load", L.mrt_last_error(None).decode())
    out = np.zeros(n, SPHERE_DTYPE)
    n2 = L.mrt_scene_load(os.fsencode(path), out.ctypes.data, len(out), C.byref(cam), C.byref(has))
    if n2 != n:
        raise MrtError(-n2 if n2 < 0 else 6, "mrt_scene_load")
    return out, (Camera._from_c(cam) if has.value else None)


def write_image(path: str, rgba: np.ndarray):
    """rgba: (H, W, 4) f32, row 0 = bottom.  .ppm / .png -> 8-bit sRGB (what the reference's surface shows), anything else -> PFM."""
    rgba = np.ascontiguousarray(rgba, np.float32)
    h, w, _ = rgba.shape
    L = _lib.load()
    fn = L.mrt_write_ppm if path.endswith(".ppm") else L.mrt_write_png if path.endswith(".png") else L.mrt_write_pfm
    st = fn(path.encode(), rgba.ctypes.data, w, h)
    if st:
        raise MrtError(st, "write_image", path)


def width_policy(op: int, workload: Sequence[int], state: Sequence[int], util: float = 0.0, rate: float = 0.0):
    """The launch-width controller's policy (csrc/width_policy.h) on synthetic input; host only.  workload = (n_tiles, n_waves,
    max_slots, spp, n_members, counter); state = (div, mult, prev_div, prev_mult, low_windows, settled, prev_rate).  Returns
    the new state (op 0: start, op 1: a window closed) or the launch share (op 2, util = frames still running)."""
    import struct
    L = _lib.load()
    w = (C.c_uint32 * 6)(*[int(x) for x in workload])
    st = list(state) if state is not None else [0, 1, 0, 1, 0, 0, 0.0]
    bits = struct.unpack("<I", struct.pack("<f", float(st[6])))[0]
    sv = (C.c_uint32 * 7)(*[int(x) for x in st[:6]], bits)
    rc = L.mrt_debug_width_policy(op, w, sv, float(util), float(rate))
    if rc:
        raise MrtError(rc, "mrt_debug_width_policy")
    if op == 2:
        return int(sv[0])
    return [int(sv[i]) for i in range(6)] + [struct.unpack("<f", struct.pack("<I", sv[6]))[0]]


# ------------------------------------------------------------------ State

class State:
    """The in-scope part of raytracer's `State` (lib.rs:206-308) on one MI355X.

    State(args, seed) ~ State::new; set_world ~ Object::new's upload; redraw() ~
    State::redraw (raytrace pass + swap + weight/shuffle update).
    """

    def __init__(self, args: Args, seed: int = 1, device: int = 0, shard: Optional[Tuple[int, int]] = None,
                 stream: Optional[int] = None):
        self._L = _lib.load()
        self._ctx = C.c_void_p()
        self.args = args.resolved()
        st = self._L.mrt_create(C.byref(args._c()), seed, device, C.byref(self._ctx))
        if st:
            raise MrtError(st, "mrt_create", self._L.mrt_last_error(None).decode())
        if shard is not None:
            self._check(self._L.mrt_set_shard(self._ctx, shard[0], shard[1]), "mrt_set_shard")
        if stream is not None:
            self._check(self._L.mrt_set_stream(self._ctx, stream), "mrt_set_stream")

    def _check(self, st, where):
        if st:
            raise MrtError(st, where, self._L.mrt_last_error(self._ctx).decode())

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.mrt_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- scene
    def set_world(self, world):
        arr = world.to_array() if isinstance(world, World) else np.ascontiguousarray(world, SPHERE_DTYPE)
        self._check(self._L.mrt_set_world(self._ctx, arr.ctypes.data, len(arr)), "mrt_set_world")

    def set_world_raw(self, w, vec4: np.ndarray, f32: np.ndarray, i32: np.ndarray):
        """w: an MrtWorld (80 bytes) or any buffer holding the reference's 64-byte raw::World."""
        wbuf = bytes(w) if not isinstance(w, (bytes, bytearray)) else bytes(w)
        vec4 = np.ascontiguousarray(vec4, np.float32).reshape(-1, 4)
        f32 = np.ascontiguousarray(f32, np.float32)
        i32 = np.ascontiguousarray(i32, np.int32)
        self._check(self._L.mrt_set_world_raw(self._ctx, wbuf, len(wbuf), vec4.ctypes.data, len(vec4), f32.ctypes.data,
                                              len(f32), i32.ctypes.data, len(i32)), "mrt_set_world_raw")

    def set_camera(self, cam: Camera):
        self._check(self._L.mrt_set_camera(self._ctx, C.byref(cam._c())), "mrt_set_camera")

    def set_seeds(self, seeds: np.ndarray):
        seeds = np.ascontiguousarray(seeds, np.uint32)
        self._check(self._L.mrt_set_seeds(self._ctx, seeds.ctypes.data, seeds.size), "mrt_set_seeds")

    def read_seeds(self) -> np.ndarray:
        rows, w = self.shard_info()[2:]
        out = np.empty((rows, w, 4), np.uint32)
        self._check(self._L.mrt_read_seeds(self._ctx, out.ctypes.data, out.size), "mrt_read_seeds")
        return out

    # -- frame loop
    def redraw(self):
        self._check(self._L.mrt_redraw(self._ctx), "mrt_redraw")

    def render(self, frames: int = 1):
        self._check(self._L.mrt_render(self._ctx, frames), "mrt_render")

    def sync(self):
        self._check(self._L.mrt_sync(self._ctx), "mrt_sync")

    def reset(self):
        self._check(self._L.mrt_reset(self._ctx), "mrt_reset")

    @property
    def locals(self) -> MrtLocals:
        out = MrtLocals()
        self._check(self._L.mrt_get_locals(self._ctx, C.byref(out)), "mrt_get_locals")
        return out

    def set_rng_shuffle(self, shuffle: Sequence[int]):
        self._check(self._L.mrt_set_rng_shuffle(self._ctx, (C.c_uint32 * 4)(*shuffle)), "mrt_set_rng_shuffle")

    def set_rng_mode(self, mode: int):
        """0 = the reference's per-pixel stream, 1 = per-sample counter-based states (extension)."""
        self._check(self._L.mrt_set_rng_mode(self._ctx, mode), "mrt_set_rng_mode")

    def set_draw_counting(self, enabled: bool):
        """False: launches without the per-lane RNG draw counter (counters()['rng_draws'] stops advancing); same images."""
        self._check(self._L.mrt_set_draw_counting(self._ctx, int(enabled)), "mrt_set_draw_counting")

    def set_samples_per_frame(self, spp: int):
        self._check(self._L.mrt_set_samples_per_frame(self._ctx, spp), "mrt_set_samples_per_frame")

    @property
    def frames_done(self) -> int:
        return int(self._L.mrt_frames_done(self._ctx))

    # -- output
    def shard_info(self) -> Tuple[int, int, int, int]:
        r, w, rows, width = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self._L.mrt_shard_info(self._ctx, C.byref(r), C.byref(w), C.byref(rows), C.byref(width)),
                    "mrt_shard_info")
        return r.value, w.value, rows.value, width.value

    def framebuffer_device_ptr(self) -> int:
        return int(self._L.mrt_framebuffer_device_ptr(self._ctx) or 0)

    def read_framebuffer(self) -> np.ndarray:
        """(H, W, 4) f32, row 0 = bottom (world == 1) or this shard's packed (local_rows, W, 4)."""
        _, world, rows, width = self.shard_info()
        shape = (self.args.height, width, 4) if world == 1 else (rows, width, 4)
        out = np.empty(shape, np.float32)
        self._check(self._L.mrt_read_framebuffer(self._ctx, out.ctypes.data, out.size), "mrt_read_framebuffer")
        return out

    def read_counters(self) -> dict:
        c = MrtCounters()
        self._check(self._L.mrt_read_counters(self._ctx, C.byref(c)), "mrt_read_counters")
        return {"samples": int(c.samples), "world_hit_calls": int(c.world_hit_calls), "rng_draws": int(c.rng_draws),
                "lane_slots": int(c.lane_slots), "member_tests": int(c.member_tests),
                "sweep_records": int(c.sweep_records)}

    def kernel_ms_history(self, n: int = 64) -> list:
        """GPU time (ms, HIP events on the launch stream) of the render kernel of the last <= n redraws."""
        buf = (C.c_float * n)()
        got = C.c_size_t()
        self._check(self._L.mrt_kernel_ms_history(self._ctx, buf, n, C.byref(got)), "mrt_kernel_ms_history")
        return [float(buf[i]) for i in range(got.value)]

    def debug_set_hierarchy(self, max_levels: int, top_target: int):
        """Tuning / test hook: depth rule of the bounding-sphere hierarchy built by the next set_world()."""
        self._check(self._L.mrt_debug_set_hierarchy(self._ctx, max_levels, top_target), "mrt_debug_set_hierarchy")

    def debug_set_sweep(self, mode: int):
        """Tuning / test hook: 0 = automatic, 1 = SGPR-fed VALU sweep, 2 = matrix-core sweep (same image either way)."""
        self._check(self._L.mrt_debug_set_sweep(self._ctx, mode), "mrt_debug_set_sweep")

    def debug_sweep_variant(self) -> int:
        """1 = SGPR-fed VALU sweep, 2 = matrix-core sweep, for the next redraw (0 before a scene is set)."""
        return int(self._L.mrt_debug_sweep_variant(self._ctx))

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._check(self._L.mrt_last_kernel_ms(self._ctx, C.byref(ms)), "mrt_last_kernel_ms")
        return float(ms.value)

    def debug_read_pixel_costs(self) -> np.ndarray:
        """(rows, W) u32: bounce-loop trips (= world_hit calls when ray_depth > 0) of every pixel in the last frame;
        full image when unsharded, else this shard's packed rows."""
        _, world, rows, width = self.shard_info()
        packed = np.empty((rows, width), np.uint32)
        self._check(self._L.mrt_debug_read_pixel_costs(self._ctx, packed.ctypes.data, packed.size), "mrt_debug_read_pixel_costs")
        return packed[:self.args.height] if world == 1 else packed

    def debug_world_hit(self, rays: np.ndarray, n_spheres: int):
        """One world_hit per ray (rays: (n, 6) f32 = origin, unit direction) through the render kernel's sweep + walk.
        Returns (hit index (n,) i32 [-1 = miss], t (n,) f32, candidates (n, n_spheres) bool = spheres that reached the root
        tests)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        n, words = len(rays), max(1, (n_spheres + 31) // 32)
        hit = np.empty((n, 2), np.int32)
        cand = np.zeros((n, words), np.uint32)
        self._check(self._L.mrt_debug_world_hit(self._ctx, rays.ctypes.data, n, hit.ctypes.data, cand.ctypes.data, words),
                    "mrt_debug_world_hit")
        bits = np.unpackbits(cand.view(np.uint8), axis=1, bitorder="little")[:, :n_spheres].astype(bool)
        return hit[:, 0].copy(), hit[:, 1].copy().view(np.float32), bits

    def debug_last_launch(self):
        """(main, pilot or None): template-argument bits of the last render launch -- 1 COUNT, 2 PILOT, 4 CTR, 8 SMALL, 16 MFMA."""
        out = (C.c_uint32 * 2)()
        self._check(self._L.mrt_debug_last_launch(self._ctx, out), "mrt_debug_last_launch")
        return int(out[0]), (None if out[1] == 0xFFFFFFFF else int(out[1]))

    def set_wait_timeout(self, seconds: float):
        """Deadline of every wait for the GPU inside the library (0 = none): a longer wait raises MrtError(MRT_ERR_STALLED)."""
        self._check(self._L.mrt_set_wait_timeout(self._ctx, float(seconds)), "mrt_set_wait_timeout")

    def get_schedule(self) -> dict:
        """The launch schedule: a frame runs on 1 / div of the persistent waves, max(2, div) x mult frames are in flight."""
        out = (C.c_uint32 * 6)()
        self._check(self._L.mrt_get_schedule(self._ctx, out), "mrt_get_schedule")
        return {"div": int(out[0]), "mult": int(out[1]), "settled": bool(out[2]), "frames_in_flight": int(out[3]),
                "last_launch_div": int(out[4]), "max_concurrent_frames": int(out[5])}

    def set_schedule_hint(self, div: int, mult: int = 1):
        """Pin the launch schedule (what an earlier run settled at); (0, 0) = measure again.  The images do not change."""
        self._check(self._L.mrt_set_schedule_hint(self._ctx, div, mult), "mrt_set_schedule_hint")

    def debug_stream_concurrency(self, streams: int = 8) -> float:
        out = C.c_float()
        self._check(self._L.mrt_debug_stream_concurrency(self._ctx, streams, C.byref(out)), "mrt_debug_stream_concurrency")
        return float(out.value)

    def debug_set_frames_in_flight(self, slots: int):
        """How many frames may be in flight, each on a side stream of its own (1..8; 0 = automatic)."""
        self._check(self._L.mrt_debug_set_frames_in_flight(self._ctx, slots), "mrt_debug_set_frames_in_flight")

    def debug_set_schedule(self, pilot_spp: int, waves_per_cu: int):
        """Before the first redraw: samples per pixel of the pilot launch, persistent waves per CU (0 = automatic)."""
        self._check(self._L.mrt_debug_set_schedule(self._ctx, pilot_spp, waves_per_cu), "mrt_debug_set_schedule")

    def debug_set_boxes(self, mode):
        """A/B switch (large scenes, whose walk tests every node's axis-aligned box): 0 / False = the boxes are opened wide and
        never reject, 1 / 2 / True = the real boxes (default); the image is the same."""
        self._check(self._L.mrt_debug_set_boxes(self._ctx, 2 if mode is True else int(mode)), "mrt_debug_set_boxes")

    def debug_arith(self, mode: int, bits_range: Sequence[int], count: int = 0, seed: int = 1):
        """mrt_debug_arith: (tested, mismatches, smallest mismatching operand) of the kernel's unscaled sqrt (mode 0, every
        bit pattern of bits_range[0..1]) / division (mode 1 / 2, `count` random pairs) against hipcc's own."""
        out = (C.c_uint64 * 3)()
        r = (C.c_uint32 * 4)(*(list(bits_range) + [0, 0, 0, 0])[:4])
        self._check(self._L.mrt_debug_arith(self._ctx, mode, r, count, seed, out), "mrt_debug_arith")
        return int(out[0]), int(out[1]), int(out[2])

    def debug_arith_pairs(self, x: np.ndarray, y: np.ndarray) -> np.ndarray:
        """mrt_debug_arith_pairs: (n, 6) u32 = bits of x / y, unscaled quotient, sqrtf(x), unscaled root, and the two guards."""
        x = np.ascontiguousarray(x, np.float32)
        y = np.ascontiguousarray(y, np.float32)
        assert x.shape == y.shape and x.ndim == 1
        out = np.zeros((len(x), 6), np.uint32)
        self._check(self._L.mrt_debug_arith_pairs(self._ctx, x.ctypes.data, y.ctypes.data, len(x), out.ctypes.data), "mrt_debug_arith_pairs")
        return out

    def debug_set_frame_batching(self, enabled):
        """A/B switch: False / 0 makes render(frames) launch every frame on its own; True / 1 automatic; 2 / 3 force the
        batch's form (2: a lane keeps its pixel for all frames of the batch, 3: frames as layers of the tile queue)."""
        self._check(self._L.mrt_debug_set_frame_batching(self._ctx, int(enabled)), "mrt_debug_set_frame_batching")

    def last_set_world_ms(self) -> float:
        """Host wall time of the last scene upload (hierarchy build + copies); one-off per scene."""
        ms = C.c_float()
        self._check(self._L.mrt_debug_last_set_world_ms(self._ctx, C.byref(ms)), "mrt_debug_last_set_world_ms")
        return float(ms.value)

    # -- multi-GPU (one process per GPU): RCCL gather on a caller-supplied ncclComm_t
    def gather_rccl(self, nccl_comm: int, root: int = 0):
        self._check(self._L.mrt_gather_rccl(self._ctx, nccl_comm, root), "mrt_gather_rccl")

    def debug_set_gather_per_band(self, enabled: bool):
        """On the root: mrt_gather uses the cross-device form of its copies (one peer copy per band) on one device too."""
        self._check(self._L.mrt_debug_set_gather_per_band(self._ctx, int(enabled)), "mrt_debug_set_gather_per_band")

    def gathered_device_ptr(self) -> int:
        return int(self._L.mrt_gathered_device_ptr(self._ctx) or 0)

    def read_gathered(self) -> np.ndarray:
        """(H, W, 4) f32, row 0 = bottom: the full frame assembled on this (root) State by the last gather."""
        out = np.empty((self.args.height, self.args.width, 4), np.float32)
        self._check(self._L.mrt_read_gathered(self._ctx, out.ctypes.data, out.size), "mrt_read_gathered")
        return out


def gather(states: Sequence[State], root: int = 0):
    """mrt_gather: one process, len(states) contexts (states[i] = shard i of n); the full frame lands on states[root]."""
    L = _lib.load()
    arr = (C.c_void_p * len(states))(*[s._ctx for s in states])
    st = L.mrt_gather(arr, len(states), root)
    if st:
        ctx = states[root]._ctx if 0 <= root < len(states) else None
        raise MrtError(st, "mrt_gather", (L.mrt_last_error(ctx) or b"").decode())


def shard_global_row(local_row: int, rank: int, world: int) -> int:
    return int(_lib.load().mrt_shard_global_row(local_row, rank, world))


def shard_local_rows(height: int, world: int) -> int:
    return int(_lib.load().mrt_shard_local_rows(height, world))


def unshard_rows(gathered: np.ndarray, height: int) -> np.ndarray:
    """Host un-permute (mrt_unshard_rows): [world, local_rows, W, 4] rank-major -> [height, W, 4]."""
    gathered = np.ascontiguousarray(gathered, np.float32)
    world, lrows, width, _ = gathered.shape
    assert lrows == shard_local_rows(height, world)
    out = np.zeros((height, width, 4), np.float32)
    st = _lib.load().mrt_unshard_rows(gathered.ctypes.data, world, width, height, out.ctypes.data)
    if st:
        raise MrtError(st, "mrt_unshard_rows")
    return out
