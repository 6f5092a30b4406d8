"""Host-side logic of the product (no GPU): Args size rule, scene packing, camera derivation,
seeding and the accumulation schedule -- each against the reference's stated behaviour and
against the oracle's independent restatement."""
import ctypes as C

import numpy as np
import pytest

from common import to_oracle_camera, to_oracle_spheres


def test_args_defaults_and_size_rule(mrt):
    a = mrt.Args()                                    # lib.rs:27-37
    assert (a.width, a.height, a.samples_per_frame, a.ray_depth, a.max_framebuffer_weight) == (0, 0, 1, 50, 1.0)
    assert (a.resolved().width, a.resolved().height) == (800, 600)       # both 0 -> default size
    r = mrt.Args(width=300).resolved()                 # lib.rs:113-134: one zero -> square
    assert (r.width, r.height) == (300, 300)
    r = mrt.Args(height=200).resolved()
    assert (r.width, r.height) == (200, 200)
    r = mrt.Args(400, 225).resolved()
    assert (r.width, r.height) == (400, 225)


def test_default_scene_is_the_references(mrt):
    s = mrt.scene_default()                            # lib.rs:687-720
    assert len(s) == 4
    assert list(s["material_ty"]) == [1, 1, 2, 2]
    assert np.allclose(s["center"], [[0, -100.5, -1], [0, 0, -1], [-1, 0, -1], [1, 0, -1]])
    assert np.allclose(s["radius"], [100, 0.5, 0.5, 0.5])
    assert np.allclose(s["albedo"][3], [0.8, 0.6, 0.2]) and np.isclose(s["param"][2], 0.3) and s["param"][3] == 1.0


def test_pack_matches_reference_indices_and_oracle(mrt, oracle):
    w, vec4, f32, i32 = mrt.pack_world(mrt.scene_default())             # lib.rs:722-799
    assert vec4.shape == (8, 4) and len(f32) == 6 and len(i32) == 8
    assert (w.spheres.center_base_idx, w.spheres.radius_base_idx, w.spheres.material_ty_base_idx,
            w.spheres.material_idx_base_idx, w.spheres.length) == (0, 0, 0, 4, 4)
    assert (w.lambertians.albedo_base_idx, w.lambertians.length) == (4, 2)
    assert (w.metals.albedo_base_idx, w.metals.fuzz_base_idx, w.metals.length) == (6, 4, 2)
    assert list(i32) == [1, 1, 2, 2, 0, 1, 0, 1]
    for scene in (mrt.scene_default(), mrt.scene_cover(1, True)[0], mrt.scene_cover(3, False)[0],
                  mrt.scene_stress(2, 12)[0]):
        w, vec4, f32, i32 = mrt.pack_world(scene)
        pw = oracle.pack_world(to_oracle_spheres(oracle, scene))
        assert bytes(w) == bytes(pw.world)
        assert np.array_equal(vec4, pw.vec4) and np.array_equal(f32, pw.f32) and np.array_equal(i32, pw.i32)


def test_pack_rejects_unknown_material(mrt):
    s = mrt.scene_default()
    s["material_ty"][1] = 7
    with pytest.raises(mrt.MrtError):
        mrt.pack_world(s)


def test_cover_scene_shape(mrt):
    glass, cam = mrt.scene_cover(1, True)
    metal, cam2 = mrt.scene_cover(1, False)
    assert 470 <= len(glass) <= 490 and len(glass) == len(metal)
    assert np.array_equal(glass["center"], metal["center"])            # same geometry, C2 vs C3 materials
    assert (glass["material_ty"] == 3).sum() > 5 and (metal["material_ty"] == 3).sum() == 0
    assert cam.defocus_angle_deg > 0 and cam2.defocus_angle_deg == 0
    assert tuple(cam.lookfrom) == (13.0, 2.0, 3.0) and cam.vfov_deg == 20.0
    big = glass[-3:]
    assert np.allclose(big["radius"], 1.0) and np.allclose(big["center"][:, 0], [0, -4, 4])
    st, _ = mrt.scene_stress(1, 100)
    assert len(st) == 10001


def test_camera_derive_matches_oracle(mrt, oracle):
    cams = [mrt.Camera(), mrt.scene_cover(1, True)[1], mrt.scene_cover(1, False)[1], mrt.scene_stress(1, 30)[1],
            mrt.Camera(1, (1, 2, 3), (-4, 0.5, 9), (0.1, 1, 0), 47.0, 1.3, 7.5)]
    for cam in cams:
        got = mrt.camera_derive(cam)
        ref = oracle.camera_derive(to_oracle_camera(oracle, cam))
        assert bytes(got) == bytes(ref)
    with pytest.raises(mrt.MrtError):
        mrt.camera_derive(mrt.Camera(1, (0, 0, 0), (0, 0, 0)))          # lookfrom == lookat


def test_schedule_and_seeds_match_oracle(mrt, oracle):
    for n in range(0, 40):
        for mw in (1.0, 0.5, 0.9):
            assert mrt.frame_weight(n, mw) == oracle.frame_weight(n, mw)
    assert mrt.frame_weight(0, 1.0) == 0.0 and mrt.frame_weight(1, 1.0) == 0.5      # lib.rs:301-304, 424
    for seed in (0, 1, 2 ** 63 + 5):
        for f in (0, 1, 2, 1000):
            assert mrt.frame_shuffle(seed, f) == oracle.frame_shuffle(seed, f)
        assert mrt.frame_shuffle(seed, 0) == [0, 0, 0, 0]                              # lib.rs:422
        one = (C.c_uint32 * 4)()
        for p in (0, 1, 12345, 1920 * 1080 - 1, 2 ** 33):
            oracle.lib().orc_pixel_seed(seed, p, one)
            assert mrt.pixel_seed(seed, p) == list(one)


def test_image_writers(mrt, tmp_path):
    img = np.zeros((3, 4, 4), np.float32)
    img[0, :, 0] = 1.0           # bottom row red
    img[2, :, 2] = 0.25          # top row blue 0.25
    p = str(tmp_path / "a.pfm")
    mrt.write_image(p, img)
    raw = open(p, "rb").read()
    assert raw.startswith(b"PF\n4 3\n-1.0\n")
    body = np.frombuffer(raw[len(b"PF\n4 3\n-1.0\n"):], np.float32).reshape(3, 4, 3)
    assert np.array_equal(body, img[..., :3])           # PFM is bottom-up like the framebuffer
    p = str(tmp_path / "a.ppm")
    mrt.write_image(p, img)
    raw = open(p, "rb").read()
    assert raw.startswith(b"P6\n4 3\n255\n")
    body = np.frombuffer(raw[len(b"P6\n4 3\n255\n"):], np.uint8).reshape(3, 4, 3)
    assert body[2, 0, 0] == 255 and body[0, 0, 2] == 137 and body[1].sum() == 0     # flipped, sRGB OETF


def test_png_writer_round_trips_through_zlib(mrt, tmp_path):
    """mrt_write_png: a valid PNG (signature, IHDR, sRGB, IDAT of stored deflate blocks, IEND, every CRC) whose pixels are the
    PPM's, for an image large enough to need several 64 KB blocks."""
    import struct
    import zlib
    rng = np.random.default_rng(1)
    h, w = 150, 200                                  # 90 KB of scanlines -> two stored blocks
    img = rng.random((h, w, 4), dtype=np.float32) * 1.2 - 0.1
    p = str(tmp_path / "a.png")
    mrt.write_image(p, img)
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        n, ty = struct.unpack(">I4s", raw[pos:pos + 8])
        data = raw[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(ty + data) & 0xFFFFFFFF, ty
        chunks.append((ty, data))
        pos += 12 + n
    assert [c[0] for c in chunks] == [b"IHDR", b"sRGB", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 2, 0, 0, 0)
    scan = np.frombuffer(zlib.decompress(chunks[2][1]), np.uint8).reshape(h, 1 + 3 * w)
    assert (scan[:, 0] == 0).all()
    pix = scan[:, 1:].reshape(h, w, 3)
    L = mrt._lib.load()
    want = np.array([[[L.mrt_srgb8(float(v)) for v in px[:3]] for px in row] for row in img[::-1][:3]], np.uint8)
    assert np.array_equal(pix[:3], want)             # top row of the file = last framebuffer row
    q = str(tmp_path / "a.ppm")
    mrt.write_image(q, img)
    ppm = open(q, "rb").read()
    body = np.frombuffer(ppm[len(f"P6\n{w} {h}\n255\n"):], np.uint8).reshape(h, w, 3)
    assert np.array_equal(body, pix)


def test_ppm_is_the_srgb_surface_encoding(mrt):
    """lib.rs:349-351 / :1133: the present pass stores linear values to the adapter's sRGB surface, i.e. the
    piecewise sRGB OETF rounded to 8 bits -- not a gamma-2 square root."""
    L = mrt._lib.load()
    known = {0.0: 0, 1.0: 255, 2.0: 255, -1.0: 0, 0.5: 188, 0.25: 137, 0.2140: 127, 0.0031308: 10, 0.001: 3, 0.01: 25, 0.18: 118}
    for lin, code in known.items():
        assert L.mrt_srgb8(lin) == code, (lin, L.mrt_srgb8(lin), code)
    assert L.mrt_srgb8(float("nan")) == 0
    # against the closed form on a sweep, and monotonic
    xs = np.linspace(0.0, 1.0, 4001, dtype=np.float32)
    got = np.array([L.mrt_srgb8(float(x)) for x in xs])
    x64 = xs.astype(np.float64)
    enc = np.where(x64 <= 0.0031308, 12.92 * x64, 1.055 * np.power(x64, 1 / 2.4) - 0.055)
    assert np.array_equal(got, np.floor(enc * 255.0 + 0.5).astype(int))
    assert (np.diff(got) >= 0).all()


def test_unshard_rows_in_cxx_matches_the_band_layout(mrt):
    """The band un-permute of the multi-GPU gather, host side in C++ (mrt_unshard_rows / mrt_shard_global_row),
    without torch or gloo: shard (r, N) holds the 8-row bands b with b % N == r, packed."""
    rng = np.random.default_rng(3)
    for height, width, world in ((27, 5, 2), (64, 3, 8), (1080, 2, 8), (9, 4, 3), (8, 1, 1), (100, 2, 7)):
        full = rng.random((height, width, 4), dtype=np.float32)
        lrows = mrt.shard_local_rows(height, world)
        nb = (height + 7) // 8
        assert lrows == ((nb + world - 1) // world) * 8
        packed = np.zeros((world, lrows, width, 4), np.float32)
        seen = np.zeros(height, int)
        for r in range(world):
            for lr in range(lrows):
                g = mrt.shard_global_row(lr, r, world)
                assert g == ((lr // 8) * world + r) * 8 + lr % 8
                if g < height:
                    packed[r, lr] = full[g]
                    seen[g] += 1
        assert (seen == 1).all()                      # every row belongs to exactly one shard
        assert np.array_equal(mrt.unshard_rows(packed, height), full)
    # agrees with the torch-side un-permute used by the one-process-per-GPU path
    import torch
    from myraytracer_amd import dist as mdist
    packed = rng.random((3, mrt.shard_local_rows(50, 3), 4, 4), dtype=np.float32)
    assert np.array_equal(mrt.unshard_rows(packed, 50), mdist.unshard(torch.from_numpy(packed), 50).numpy())


def test_scene_files_round_trip_every_bit(mrt, tmp_path):
    """Scenes as data: save -> load reproduces every sphere field and the camera bit for bit."""
    cases = [(mrt.scene_default(), None), mrt.scene_cover(1, True), mrt.scene_cover(7, False), mrt.scene_stress(2, 12)]
    odd = np.zeros(4, mrt.SPHERE_DTYPE)
    odd[0] = ((1e-30, -3.4e38, 0.1), -0.5, 1, (0.1, 0.2, 0.3), -0.0)        # Lambertian with a non-canonical param
    odd[1] = ((1 / 3, 2 / 3, 1e6), 1e-3, 3, (0.5, 1.0, 1.0), 1.33)          # Dielectric with a colour
    odd[2] = ((0, 0, 0), 1.0, 7, (0.25, 0.5, 0.75), 9.0)                    # unknown MaterialTy: absorbs
    odd[3] = ((1, 2, 3), 4.0, 2, (0.9, 0.8, 0.7), 0.123456789)
    cases.append((odd, mrt.Camera(1, (0.1, 0.2, 0.3), (1, 1, 1), (0, 1, 0), 33.3, 0.7, 9.99)))
    for i, (sc, cam) in enumerate(cases):
        path = str(tmp_path / f"scene{i}.txt")
        mrt.save_scene(path, sc, cam)
        sc2, cam2 = mrt.load_scene(path)
        assert sc2.tobytes() == np.ascontiguousarray(sc, mrt.SPHERE_DTYPE).tobytes()
        assert (cam2 is None) == (cam is None)
        if cam is not None:
            assert bytes(cam2._c()) == bytes(cam._c())


def test_scene_file_errors_name_the_line(mrt, tmp_path):
    bad = tmp_path / "bad.txt"
    bad.write_text("# ok\nsphere 0 0 0 1 lambertian 0.5 0.5 0.5\nsphere 0 0 0 1 velvet 1 2 3\n")
    with pytest.raises(mrt.MrtError) as e:
        mrt.load_scene(str(bad))
    assert "line 3" in str(e.value)
    with pytest.raises(mrt.MrtError):
        mrt.load_scene(str(tmp_path / "missing.txt"))
    ok = tmp_path / "hand.txt"
    ok.write_text("camera pinhole\n  sphere 0 -100.5 -1 100 lambertian 0.8 0.8 0   # ground\n\nsphere 1 0 -1 .5 metal .8 .6 .2 1\n")
    sc, cam = mrt.load_scene(str(ok))
    assert len(sc) == 2 and cam is not None and cam.mode == 0
    assert sc[1]["material_ty"] == mrt.METAL and sc[1]["param"] == 1.0 and tuple(sc[0]["center"]) == (0.0, -100.5, -1.0)


def test_matrix_core_sweep_scale_keeps_the_clamp_from_saturating(mrt):
    """api.cpp mfma_scales: the sweep squares g = K oc.ds with an instruction that clamps to [0, 1] (kernels.hip,
    mfma_sweep_tile), so K must be a power of two (no rounding changes) with |g| <= 1/2 for every admitted ray: origins up
    to 4 x reach, records within reach, |ds| < 1.001."""
    import ctypes as C
    import struct
    from myraytracer_amd import _lib
    L = _lib.load()
    for reach in [1e-4, 0.03, 0.5, 1.0, 13.7, 1000.0, 1.73e7, 6.0e7]:
        sc = (C.c_float * 4)()
        pair = C.c_uint32()
        assert L.mrt_debug_mfma_scale(reach, sc, C.byref(pair)) == 0
        K2 = sc[1] / 2.0
        K = K2 ** 0.5
        m, e = np.frexp(K)
        assert m == 0.5, "K is a power of two"
        assert 5.01 * reach * K <= 0.5 and 5.01 * reach * K > 0.0624          # |g| <= 1/2, and no more than 3 bits given away
        assert sc[0] == np.float32(np.float32(1.0001) * np.float32(K))         # kBoundStretch x K, exact scaling
        assert sc[2] == np.float32(-(1.0 - 2.0 ** -13) * K2)
        assert sc[3] == np.float32(16.0 * reach * reach)
        lo, hi = pair.value & 0xFFFF, pair.value >> 16
        assert lo == hi and struct.unpack("<f", struct.pack("<I", lo << 16))[0] == -K2   # -K^2 as bf16, exact
    sc = (C.c_float * 4)()
    pair = C.c_uint32()
    assert L.mrt_debug_mfma_scale(float("nan"), sc, C.byref(pair)) != 0
    assert L.mrt_debug_mfma_scale(0.0, sc, C.byref(pair)) == 0 and sc[1] > 0      # an empty scene: any K
