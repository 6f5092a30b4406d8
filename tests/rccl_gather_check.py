#!/usr/bin/env python3
"""Run by tests/test_gpu_multi.py in a FRESH process (so that exactly one librccl is in it): drives
mrt_gather_rccl on an RCCL communicator created by this caller with /opt/rocm's librccl.

A one-GPU box allows a world of one (RCCL refuses two ranks on one device), which still exercises the
run-time binding of RCCL, the communicator checks, the root's staging / un-permute and mrt_read_gathered;
the N > 1 exchange itself (grouped ncclSend / ncclRecv) is what bench.py --gpus N exercises on the 8-GPU node
through torch.distributed, with the same band layout.  Prints "ok" and exits 0 on success.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MRT_HIP_RUNTIME"] = "system"      # a process without torch: /opt/rocm's HIP runtime for the library and for librccl
import myraytracer_amd as M  # noqa: E402


def main():
    rccl = C.CDLL(os.environ.get("MRT_RCCL_LIB", "/opt/rocm/lib/librccl.so.1"), mode=C.RTLD_GLOBAL)
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    rccl.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    rc = rccl.ncclCommInitAll(C.byref(comm), 1, devs)
    assert rc == 0, f"ncclCommInitAll -> {rc}"
    sc, cam = M.scene_cover(1, True)
    with M.State(M.Args(96, 54, 2, 50), seed=1) as st:
        st.set_world(sc)
        st.set_camera(cam)
        st.redraw()
        st.gather_rccl(comm.value, 0)
        whole = st.read_gathered()
        assert np.array_equal(whole.view(np.uint32), st.read_framebuffer().view(np.uint32))
        # a communicator that does not match the shard is refused, not used
        try:
            with M.State(M.Args(96, 54, 2, 50), seed=1, shard=(1, 2)) as other:
                other.set_world(sc)
                other.redraw()
                other.gather_rccl(comm.value, 0)
            raise SystemExit("a world-1 communicator was accepted for shard 1 of 2")
        except M.MrtError as e:
            assert e.status == 7, e
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
    print("ok")


if __name__ == "__main__":
    main()
