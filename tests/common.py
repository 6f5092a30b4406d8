"""Shared helpers for the parity tests: the same scene fed to the HIP path and to the oracle."""
import numpy as np


def to_oracle_spheres(O, spheres):
    """Product AoS (center, radius, material_ty, albedo, param) and the oracle's AoS
    (center, radius, ty, p[4]) have the same 36-byte layout."""
    return np.ascontiguousarray(spheres).view(O.SPHERE_DTYPE)


def to_oracle_camera(O, cam):
    if cam is None or cam.mode == 0:
        return O.pinhole_camera()
    return O.lookat_camera(cam.lookfrom, cam.lookat, cam.vup, cam.vfov_deg, cam.defocus_angle_deg, cam.focus_dist)


def oracle_render(O, spheres, cam, width, height, spp, depth, seed, frames=1, max_w=1.0, counters=None, rng_mode=0):
    packed = O.pack_world(to_oracle_spheres(O, spheres))
    return O.render(width, height, spp, depth, packed, to_oracle_camera(O, cam), seed, frames=frames,
                    max_w=max_w, counters=counters, rng_mode=rng_mode)


def gpu_render(M, spheres, cam, width, height, spp, depth, seed, frames=1, max_w=1.0, shard=None, rng_mode=0):
    args = M.Args(width, height, spp, depth, max_w)
    with M.State(args, seed=seed, shard=shard) as st:
        st.set_world(spheres)
        if cam is not None:
            st.set_camera(cam)
        if rng_mode:
            st.set_rng_mode(rng_mode)
        st.render(frames)
        st.sync()
        return st.read_framebuffer(), st.read_counters(), st.last_kernel_ms()


def rmse_rgb(a, b):
    d = a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)
    return float(np.sqrt(np.mean(d * d)))


def mismatch_report(a, b):
    neq = (a.view(np.uint32) != b.view(np.uint32)).any(axis=-1)
    n = int(neq.sum())
    if n == 0:
        return "bit-identical"
    ys, xs = np.nonzero(neq)
    return (f"{n} of {neq.size} pixels differ; first at (x={xs[0]}, y={ys[0]}): gpu={a[ys[0], xs[0]]} "
            f"oracle={b[ys[0], xs[0]]}; rmse={rmse_rgb(a, b):.3e}")
