"""myraytracer_amd: MI355X-native backend for the per-pixel render loop of
zetanumbers/myraytracer (raytracer/src/shader.wgsl), behind a C ABI.

Importing this package loads myraytracer_amd/lib/libmyraytracer_amd.so and fails loudly
if it has not been built; rendering additionally requires a gfx950 GPU.
"""
from . import _lib
from .api import (DIELECTRIC, LAMBERTIAN, METAL, SPHERE_DTYPE, Args, Camera, Dielectric, Lambertian, Metal,
                  MrtError, Sphere, State, World, camera_derive, frame_shuffle, frame_weight, pack_world,
                  gather, load_scene, pixel_seed, save_scene, scene_cover, scene_default, scene_stress, shard_global_row,
                  shard_local_rows, unshard_rows, width_policy, write_image)

_lib.load()

__all__ = ["Args", "Camera", "Dielectric", "Lambertian", "Metal", "Sphere", "State", "World", "MrtError",
           "LAMBERTIAN", "METAL", "DIELECTRIC", "SPHERE_DTYPE", "pack_world", "camera_derive", "frame_weight",
           "frame_shuffle", "pixel_seed", "scene_default", "scene_cover", "scene_stress", "save_scene", "load_scene", "write_image",
           "gather", "shard_global_row", "shard_local_rows", "unshard_rows", "width_policy"]
