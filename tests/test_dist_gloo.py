"""Multi-process path on CPU: world_size-2 gloo.  Each rank produces its interleaved 8-row
bands (the oracle stands in for the GPU renderer -- tests only), the product's gather +
un-permute (myraytracer_amd.dist) must rebuild the single-process image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import myraytracer_amd as M
    from myraytracer_amd import dist as mdist
    from oracle import pyoracle as O
    from common import to_oracle_spheres
    spheres = M.scene_default()
    packed = O.pack_world(to_oracle_spheres(O, spheres))
    seeds = O.fill_seeds(5, width, height)
    lrows = mdist.local_rows(height, world)
    local = np.zeros((lrows, width, 4), np.float32)
    for lr in range(0, lrows, mdist.BAND_ROWS):
        y0 = mdist.global_row(lr, rank, world)
        if y0 >= height:
            continue
        y1 = min(height, y0 + mdist.BAND_ROWS)
        full = O.render_frame(width, height, 2, 8, packed, O.pinhole_camera(), seeds, rows=(y0, y1), nthreads=2)
        local[lr:lr + (y1 - y0)] = full[y0:y1]
    img = mdist.gather_framebuffer(torch.from_numpy(local), height, dst=0)
    if rank == 0:
        np.save(out_path, img.numpy())
    else:
        assert img is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 36), (2, 45), (3, 20)])
def test_gather_unshard_world(tmp_path, oracle, mrt, world, height):
    width = 40
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), width, height, out), nprocs=world, join=True)
    got = np.load(out)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from common import to_oracle_spheres
    packed = oracle.pack_world(to_oracle_spheres(oracle, mrt.scene_default()))
    ref = oracle.render_frame(width, height, 2, 8, packed, oracle.pinhole_camera(), oracle.fill_seeds(5, width, height))
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


class _FakeState:
    """What share_schedule needs of a State (get_schedule / set_schedule_hint), without a GPU."""

    def __init__(self, div, mult, settled):
        self.sch = {"div": div, "mult": mult, "settled": settled}
        self.pinned = None

    def get_schedule(self):
        return dict(self.sch)

    def set_schedule_hint(self, div, mult):
        self.pinned = (div, mult)
        self.sch = {"div": div, "mult": mult, "settled": True}


def _schedule_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from myraytracer_amd import dist as mdist
    # rank 0 settled at a quarter width, eight in flight; rank 1's controller went elsewhere; rank 2 has not settled
    st = [_FakeState(4, 2, True), _FakeState(8, 2, True), _FakeState(4, 2, False)][rank]
    got = mdist.share_schedule(st, 0)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([got[0], got[1], -1 if st.pinned is None else st.pinned[0]]))
    # a root that has no setting yet: nobody is pinned
    st2 = [_FakeState(0, 1, False), _FakeState(8, 2, True), _FakeState(2, 2, True)][rank]
    assert mdist.share_schedule(st2, 0) is None and st2.pinned is None
    dist.barrier()
    dist.destroy_process_group()


def test_share_schedule_gives_every_rank_the_roots_setting(tmp_path, mrt):
    """N > 1: the ranks render equal shares of one frame, and each library instance measures its schedule on its own host
    clock; bench.py (and any caller) broadcasts rank 0's and pins it (mrt_set_schedule_hint) -- world 3 over gloo."""
    world = 3
    mp.spawn(_schedule_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(str(tmp_path / f"r{k}.npy")) for k in range(world)]
    assert [tuple(x[:2]) for x in r] == [(4, 2)] * 3
    assert r[0][2] == -1            # the root keeps what it has
    assert r[1][2] == 4             # a rank that had settled elsewhere is pinned to the root's
    assert r[2][2] == 4             # so is a rank that had not settled yet


def test_band_layout_math(mrt):
    from myraytracer_amd import dist as mdist
    assert mdist.band_layout(1080, 1) == (135, 135)
    assert mdist.band_layout(1080, 8) == (135, 17)
    assert mdist.band_layout(2160, 8) == (270, 34)
    assert mdist.local_rows(45, 2) == 24
    seen = set()
    for rank in range(3):
        for lr in range(mdist.local_rows(100, 3)):
            g = mdist.global_row(lr, rank, 3)
            assert g not in seen
            seen.add(g)
    assert set(range(100)) <= seen
    # unshard is the inverse of the packing
    world, h, w = 3, 100, 5
    full = torch.arange(h * w * 4, dtype=torch.float32).reshape(h, w, 4)
    lrows = mdist.local_rows(h, world)
    packed = torch.zeros(world, lrows, w, 4)
    for rank in range(world):
        for lr in range(lrows):
            g = mdist.global_row(lr, rank, world)
            if g < h:
                packed[rank, lr] = full[g]
    assert torch.equal(mdist.unshard(packed, h), full)
