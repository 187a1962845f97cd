"""lib/flow_c Python entry — mirror of lib/flow_c/gpu_flow.pyx:24-41 (`gpu_flow`) and
lib/flow_c/flow.py:19-23 (`gpu_flow_wrapper`), bound with ctypes to the `_flow` C symbol
(same signature as lib/flow_c/gpu_flow.hpp:1-3) instead of a Cython extension."""
import ctypes

import numpy as np

from ...runtime import lib


def gpu_flow(depth_src, depth_tgt, KT, Kinv, device_id=0):
    """depth_src, depth_tgt (n,1,h,w), KT (n,3,4), Kinv (3,3) float32 → flow (n,2,h,w), valid (n,1,h,w)."""
    depth_src = np.ascontiguousarray(depth_src, dtype=np.float32)
    depth_tgt = np.ascontiguousarray(depth_tgt, dtype=np.float32)
    KT = np.ascontiguousarray(KT, dtype=np.float32)
    Kinv = np.ascontiguousarray(Kinv, dtype=np.float32)
    batch_size, _, height, width = depth_src.shape
    assert depth_tgt.shape == depth_src.shape and KT.shape == (batch_size, 3, 4) and Kinv.shape == (3, 3)
    flow = np.zeros((batch_size, 2, height, width), dtype=np.float32)
    valid = np.zeros((batch_size, 1, height, width), dtype=np.float32)
    dll = lib.load()
    p = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731
    dll._flow(p(flow), p(valid), p(depth_src), p(depth_tgt), p(KT), p(Kinv), batch_size, height, width, int(device_id))
    if dll.deepim_flow_status() != 0:
        raise RuntimeError("_flow failed: %s" % lib.last_error())
    return flow, valid


def gpu_flow_wrapper(device_id):
    def _flow(depth_src, depth_tgt, KT, Kinv):
        return gpu_flow(depth_src, depth_tgt, KT, Kinv, device_id)

    return _flow
