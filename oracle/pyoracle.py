"""ctypes loader for the CPU ORACLE (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package myraytracer_amd.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")


class SphereRange(C.Structure):
    _fields_ = [("center_base_idx", C.c_int32), ("radius_base_idx", C.c_int32),
                ("material_ty_base_idx", C.c_int32), ("material_idx_base_idx", C.c_int32),
                ("length", C.c_int32), ("_pad", C.c_int32 * 3)]


class LambertianRange(C.Structure):
    _fields_ = [("albedo_base_idx", C.c_int32), ("length", C.c_int32), ("_pad", C.c_int32 * 2)]


class MetalRange(C.Structure):
    _fields_ = [("albedo_base_idx", C.c_int32), ("fuzz_base_idx", C.c_int32),
                ("length", C.c_int32), ("_pad", C.c_int32)]


class DielectricRange(C.Structure):
    _fields_ = [("ior_base_idx", C.c_int32), ("length", C.c_int32), ("_pad", C.c_int32 * 2)]


class World(C.Structure):
    _fields_ = [("spheres", SphereRange), ("lambertians", LambertianRange),
                ("metals", MetalRange), ("dielectrics", DielectricRange)]


class Locals(C.Structure):
    _fields_ = [("shape", C.c_uint32 * 2), ("samples_per_frame", C.c_uint32),
                ("ray_depth", C.c_uint32), ("rng_shuffle", C.c_uint32 * 4),
                ("framebuffer_weight", C.c_float), ("rng_mode", C.c_uint32), ("_pad", C.c_uint32 * 2)]


class Camera(C.Structure):
    _fields_ = [("mode", C.c_int32), ("lookfrom", C.c_float * 3), ("lookat", C.c_float * 3),
                ("vup", C.c_float * 3), ("vfov_deg", C.c_float),
                ("defocus_angle_deg", C.c_float), ("focus_dist", C.c_float)]


class CameraRaw(C.Structure):
    _fields_ = [("mode", C.c_int32), ("defocus", C.c_int32), ("origin", C.c_float * 3),
                ("su", C.c_float * 3), ("sv", C.c_float * 3), ("fw", C.c_float * 3),
                ("ru", C.c_float * 3), ("rv", C.c_float * 3)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "samples", "world_hit_calls", "sphere_tests", "rng_draws",
        "scatter_lambertian", "scatter_metal", "scatter_dielectric",
        "paths_missed", "paths_absorbed", "paths_exhausted")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Hit(C.Structure):
    _fields_ = [("at", C.c_float * 3), ("t", C.c_float), ("normal", C.c_float * 3),
                ("front_face", C.c_int32), ("ty", C.c_int32), ("idx", C.c_int32)]


class SphereAoS(C.Structure):
    _fields_ = [("center", C.c_float * 3), ("radius", C.c_float), ("ty", C.c_int32), ("p", C.c_float * 4)]


SPHERE_DTYPE = np.dtype([("center", "<f4", 3), ("radius", "<f4"), ("ty", "<i4"), ("p", "<f4", 4)])
LAMBERTIAN, METAL, DIELECTRIC = 1, 2, 3

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_readings = {}


class reading:
    """with pyoracle.reading("nofma"): ... -- every call inside runs the oracle built with the OTHER legal reading of the
    shader's arithmetic (no fused multiply-add anywhere; oracle/Makefile, rt_oracle.c ORC_READING_NOFMA).  Only for
    scripts/reading_spread.py / tests/test_reading_spread.py: it is never what the GPU path is compared with."""

    def __init__(self, name):
        assert name == "nofma"
        self.path = os.path.join(_HERE, "librt_oracle_nofma.so")

    def __enter__(self):
        global _lib
        self.saved = lib()
        if self.path not in _readings:
            if not os.path.exists(self.path):
                build()
            _lib = None
            _readings[self.path] = lib(self.path)
        _lib = _readings[self.path]
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved


def lib(path=None):
    global _lib
    if _lib is None:
        if not os.path.exists(path or _LIB_PATH):
            build()
        L = C.CDLL(path or _LIB_PATH)
        L.orc_xoshiro128plus_next.restype = C.c_uint32
        L.orc_xoshiro128plus_next.argtypes = [C.POINTER(C.c_uint32)]
        L.orc_u32_to_f32.restype = C.c_float
        L.orc_u32_to_f32.argtypes = [C.c_uint32]
        L.orc_splitmix64_at.restype = C.c_uint64
        L.orc_splitmix64_at.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_pixel_seed.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]
        L.orc_fill_seeds.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_frame_shuffle.argtypes = [C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_frame_weight.restype = C.c_float
        L.orc_frame_weight.argtypes = [C.c_uint32, C.c_float]
        L.orc_sphere_hit.restype = C.c_int
        L.orc_sphere_hit.argtypes = [C.POINTER(World), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                                     C.POINTER(Hit)]
        L.orc_world_hit.restype = C.c_int
        L.orc_world_hit.argtypes = [C.POINTER(World), C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                                    C.POINTER(Hit), C.POINTER(C.c_int32)]
        L.orc_color_sky.argtypes = [C.c_float, C.POINTER(C.c_float)]
        L.orc_camera_derive.argtypes = [C.POINTER(Camera), C.POINTER(CameraRaw)]
        L.orc_pack_world.argtypes = [C.c_void_p, C.c_int32, C.POINTER(World), C.c_void_p,
                                     C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32),
                                     C.c_void_p, C.POINTER(C.c_int32)]
        L.orc_render_rows.argtypes = [C.POINTER(Locals), C.POINTER(World), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.POINTER(CameraRaw), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(Counters)]
        L.orc_render_rect.argtypes = [C.POINTER(Locals), C.POINTER(World), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.POINTER(CameraRaw), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                      C.POINTER(Counters)]
        L.orc_world_hit_batch.argtypes = [C.POINTER(World), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class PackedWorld:
    """raw::World + the three SoA arrays (lib.rs:641-685, 722-799)."""

    def __init__(self, world, vec4, f32, i32):
        self.world, self.vec4, self.f32, self.i32 = world, vec4, f32, i32


def pack_world(spheres):
    """spheres: numpy structured array of SPHERE_DTYPE (AoS, mirrors api::Sphere lib.rs:611-639)."""
    spheres = np.ascontiguousarray(spheres, dtype=SPHERE_DTYPE)
    n = len(spheres)
    vec4 = np.zeros((2 * n + 1, 4), np.float32)
    f32 = np.zeros(2 * n + 1, np.float32)
    i32 = np.zeros(2 * n + 1, np.int32)
    w = World()
    nv, nf, ni = C.c_int32(), C.c_int32(), C.c_int32()
    lib().orc_pack_world(_ptr(spheres), n, C.byref(w), _ptr(vec4), C.byref(nv), _ptr(f32), C.byref(nf),
                         _ptr(i32), C.byref(ni))
    return PackedWorld(w, vec4[:nv.value].copy(), f32[:nf.value].copy(), i32[:ni.value].copy())


def world_hit_batch(packed, rays, nthreads=0):
    """world_hit (shader.wgsl:314-329, range [0.001, 1e4)) per ray -> (winner index or -1, t, disc >= 0 matrix [n, spheres],
    the same restricted to spheres not entirely behind the ray's origin)."""
    rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
    n, ns = len(rays), int(packed.world.spheres.length)
    hit = np.empty(n, np.int32)
    t = np.empty(n, np.float32)
    ge0 = np.zeros((n, ns), np.uint8)
    lib().orc_world_hit_batch(C.byref(packed.world), _ptr(packed.vec4), _ptr(packed.f32), _ptr(packed.i32), _ptr(rays), n,
                              _ptr(hit), _ptr(t), _ptr(ge0), nthreads)
    return hit, t, (ge0 & 1).astype(bool), (ge0 & 2).astype(bool)


def fill_seeds(seed, w, h):
    out = np.empty((h, w, 4), np.uint32)
    lib().orc_fill_seeds(seed, w, h, _ptr(out))
    return out


def frame_shuffle(seed, frame):
    out = (C.c_uint32 * 4)()
    lib().orc_frame_shuffle(seed, frame, out)
    return [int(x) for x in out]


def frame_weight(frames_done, max_w):
    return float(lib().orc_frame_weight(frames_done, max_w))


def camera_derive(cam):
    raw = CameraRaw()
    lib().orc_camera_derive(C.byref(cam), C.byref(raw))
    return raw


def pinhole_camera():
    c = Camera()
    c.mode = 0
    return c


def lookat_camera(lookfrom, lookat, vup, vfov, defocus_angle, focus_dist):
    c = Camera()
    c.mode = 1
    c.lookfrom[:] = lookfrom
    c.lookat[:] = lookat
    c.vup[:] = vup
    c.vfov_deg, c.defocus_angle_deg, c.focus_dist = vfov, defocus_angle, focus_dist
    return c


def render_frame(width, height, spp, depth, packed, cam, seeds, shuffle=(0, 0, 0, 0), weight=0.0,
                 prev=None, rows=None, nthreads=0, counters=None, rng_mode=0, cols=None):
    """One pass of fs_main (shader.wgsl:371-386) over rows [rows[0], rows[1]) (and columns [cols[0], cols[1]))
    -> (H,W,4) f32, row 0 = bottom; texels outside the rectangle stay 0."""
    L = Locals()
    L.shape[0], L.shape[1] = width, height
    L.samples_per_frame, L.ray_depth = spp, depth
    L.rng_shuffle[:] = list(shuffle)
    L.framebuffer_weight = weight
    L.rng_mode = rng_mode
    if prev is None:
        prev = np.zeros((height, width, 4), np.float32)
    prev = np.ascontiguousarray(prev, np.float32)
    seeds = np.ascontiguousarray(seeds, np.uint32)
    assert seeds.shape == (height, width, 4) and prev.shape == (height, width, 4)
    out = np.zeros((height, width, 4), np.float32)
    y0, y1 = (0, height) if rows is None else rows
    raw = camera_derive(cam) if isinstance(cam, Camera) else cam
    x0, x1 = (0, width) if cols is None else cols
    lib().orc_render_rect(C.byref(L), C.byref(packed.world), _ptr(packed.vec4), _ptr(packed.f32),
                          _ptr(packed.i32), C.byref(raw), _ptr(seeds), _ptr(prev), _ptr(out),
                          x0, x1, y0, y1, nthreads, C.byref(counters) if counters is not None else None)
    return out


def render(width, height, spp, depth, packed, cam, seed, frames=1, max_w=1.0, nthreads=0, counters=None, rng_mode=0):
    """The progressive loop of State::redraw (lib.rs:241-307): `frames` frames of `spp` samples."""
    seeds = fill_seeds(seed, width, height)
    fb = np.zeros((height, width, 4), np.float32)
    for f in range(frames):
        fb = render_frame(width, height, spp, depth, packed, cam, seeds, frame_shuffle(seed, f),
                          frame_weight(f, max_w), fb, None, nthreads, counters, rng_mode)
    return fb
