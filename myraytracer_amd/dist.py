"""Multi-GPU tile sharding of the render path: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The reference is single-device; sharding is this build's addition.  Pixels
are independent (own RNG stream keyed by global pixel index, own output texel; the scene is
replicated), so the only exchange is ONE gather of the RGBA32F bands to the root at the end
of a frame.  Bands of 8 rows are dealt round-robin (band b -> rank b % world) because cost
per row is very uneven (sky rows end at bounce 0).  Each of the 7 peers has its own xGMI
link to the root, so a direct gather (grouped send/recv) uses all links at once; there is
nothing to reduce, hence no ring collective.
"""
from typing import Optional, Tuple

import torch
import torch.distributed as dist

BAND_ROWS = 8     # mrt::kBandRows


def band_layout(height: int, world: int) -> Tuple[int, int]:
    """(total bands, bands per rank); every rank holds the same count (the tail is padding)."""
    nb = (height + BAND_ROWS - 1) // BAND_ROWS
    return nb, (nb + world - 1) // world


def local_rows(height: int, world: int) -> int:
    return band_layout(height, world)[1] * BAND_ROWS


def global_row(local_row: int, rank: int, world: int) -> int:
    """Inverse of the packing documented at mrt_shard_info."""
    return ((local_row // BAND_ROWS) * world + rank) * BAND_ROWS + local_row % BAND_ROWS


def unshard(gathered: torch.Tensor, height: int) -> torch.Tensor:
    """gathered: [world, local_rows, W, 4] (rank-major) -> [height, W, 4], row 0 = bottom."""
    world, lrows, width, ch = gathered.shape
    nbl = lrows // BAND_ROWS
    full = gathered.reshape(world, nbl, BAND_ROWS, width, ch).permute(1, 0, 2, 3, 4)
    return full.reshape(nbl * world * BAND_ROWS, width, ch)[:height].contiguous()


def gather_framebuffer(local: torch.Tensor, height: int, dst: int = 0,
                       out: Optional[torch.Tensor] = None, group=None) -> Optional[torch.Tensor]:
    """Gather every rank's packed bands to `dst` and un-permute them into the full image.

    local: [local_rows, W, 4] f32 on this rank's device.  `out` (dst only) may be a
    preallocated [world, local_rows, W, 4] staging tensor.  Returns the image on dst, None elsewhere.
    """
    if not dist.is_initialized():
        return unshard(local.unsqueeze(0), height)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        if out is None:
            out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.gather(local, list(out.unbind(0)), dst=dst, group=group)
        return unshard(out, height)
    dist.gather(local, None, dst=dst, group=group)
    return None


def share_schedule(state, src: int = 0, group=None):
    """One launch schedule for every rank: broadcast rank `src`'s setting (mrt_get_schedule) and pin it everywhere
    (mrt_set_schedule_hint).  The library measures its setting on each host's clock, so the ranks of a run -- equal shares of one
    frame -- may otherwise settle differently.  Collective; returns the (div, mult) now in force, or None if `src` has none yet."""
    mine = state.get_schedule()
    box = [(mine["div"], mine["mult"]) if mine["div"] else None]
    if dist.is_initialized():
        dist.broadcast_object_list(box, src=src, group=group)
    if box[0] is None:
        return None
    if box[0] != (mine["div"], mine["mult"]) or not mine["settled"]:
        state.set_schedule_hint(*box[0])
    return box[0]


class _DevicePtr:
    """Wraps a raw device pointer owned by an mrt_ctx as a __cuda_array_interface__ object."""

    def __init__(self, ptr: int, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (ptr, False),
                                         "version": 2, "strides": None}


def framebuffer_tensor(state, device: Optional[torch.device] = None) -> torch.Tensor:
    """Zero-copy torch view of the State's most recent framebuffer ([local_rows, W, 4] f32).

    Stream ordering is the caller's: torch ops on this tensor follow the State's redraw only if the State was created on
    the stream they run on -- `State(..., stream=s.cuda_stream)` with a real `torch.cuda.Stream` s made current (torch's
    default stream has handle 0, which the ABI reads as "use your own stream") -- or after `state.sync()`."""
    _, _, rows, width = state.shard_info()
    ptr = state.framebuffer_device_ptr()
    if not ptr:
        raise RuntimeError("no framebuffer")
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.as_tensor(_DevicePtr(ptr, (rows, width, 4)), device=dev)


def gathered_tensor(root_state, device: Optional[torch.device] = None) -> torch.Tensor:
    """Zero-copy torch view of the full frame the last mrt_gather / mrt_gather_rccl assembled on this root State
    ([height, W, 4] f32, row 0 = bottom).  Valid until the next gather; ops on it must run on the State's stream (the next
    gather's copies wait for what that stream holds, see mrt_gathered_device_ptr)."""
    ptr = root_state.gathered_device_ptr()
    if not ptr:
        raise RuntimeError("nothing gathered yet")
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.as_tensor(_DevicePtr(ptr, (root_state.args.height, root_state.args.width, 4)), device=dev)
