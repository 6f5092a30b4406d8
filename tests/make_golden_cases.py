"""Case table + renderer shared by tests/golden/make_golden.py and the golden tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CASES = [
    # name, scene, width, height, spp, depth, seed, frames, max_w
    dict(name="c1_default_80x45", scene="default", width=80, height=45, spp=16, depth=8, seed=1, frames=1, max_w=1.0),
    dict(name="c2_cover_metal_96x54", scene="cover", width=96, height=54, spp=4, depth=50, seed=1, frames=1, max_w=1.0),
    dict(name="c3_cover_glass_96x54", scene="cover-glass", width=96, height=54, spp=2, depth=50, seed=1, frames=2,
         max_w=1.0),
    dict(name="ema_default_48x27", scene="default", width=48, height=27, spp=1, depth=8, seed=9, frames=4, max_w=0.6),
]


def load_inputs(case):
    """Inputs are stored with the fixture: spheres (36-byte AoS records) and the camera."""
    raw = np.fromfile(os.path.join(GOLDEN, case["spheres_file"]), np.uint8)
    cam = case["camera"]
    return raw, cam


def render_case(O, case):
    raw, cam = load_inputs(case)
    spheres = raw.view(O.SPHERE_DTYPE)
    if cam["mode"] == 0:
        ocam = O.pinhole_camera()
    else:
        ocam = O.lookat_camera(cam["lookfrom"], cam["lookat"], cam["vup"], cam["vfov_deg"], cam["defocus_angle_deg"],
                               cam["focus_dist"])
    cnt = O.Counters()
    fb = O.render(case["width"], case["height"], case["spp"], case["depth"], O.pack_world(spheres), ocam, case["seed"],
                  frames=case["frames"], max_w=case["max_w"], counters=cnt)
    return fb, cnt.as_dict()
