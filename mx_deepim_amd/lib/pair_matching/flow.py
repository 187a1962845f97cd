"""`calc_flow` on the device — mirror of lib/pair_matching/flow.py:12-63 (float64 numpy variant used by
the loader and the EPE evaluation).  Returns (flow (H,W,2) float32, visible (H,W) float32)."""
import ctypes

import numpy as np

from ...runtime import Context, lib
from .RT_transform import calc_KT


def calc_flow(depth_src, pose_src, pose_tgt, K, depth_tgt, thresh=3e-3, standard_rep=False, ctx=None):
    ctx = ctx or Context.default()
    depth_src = np.ascontiguousarray(depth_src, np.float32)
    depth_tgt = np.ascontiguousarray(depth_tgt, np.float32)
    H, W = depth_src.shape[:2]
    K = np.ascontiguousarray(K, np.float32).reshape(3, 3)
    Kinv = np.ascontiguousarray(np.linalg.inv(K))  # float32 in, float32 out — as np.linalg.inv(np.matrix(K)) gives
    KT = calc_KT(np.asarray(pose_src, np.float32)[None], np.asarray(pose_tgt, np.float32)[None], K, ctx)
    flow, vis = ctx.empty((1, H, W, 2)), ctx.empty((1, H, W))
    lib.deepim_calc_flow_forward(ctx.handle, flow, vis, ctx.array(depth_src), ctx.array(depth_tgt), KT, Kinv,
                                 ctypes.c_float(thresh), 1 if standard_rep else 0, 1, H, W)
    return flow.asnumpy()[0], vis.asnumpy()[0]
