"""Device-side pose update — mirror of the hot-path entry points of lib/pair_matching/RT_transform.py:
`RT_transform` (:127-151, quaternion form) and `calc_se3`-based `K·T` (batch_updater_py_multi.py:255-259).
The algebra runs in the rt_transform / calc_KT HIP kernels with the reference's float32/float64 pattern."""
import numpy as np

from ...config import ROT_COORD_CODE
from ...runtime import Context, DeviceArray, lib


def RT_transform_batch(pose_src, se3, T_means, T_stds, rot_coord="MODEL", out=None, ctx=None):
    """pose_src (B,3,4), se3 (B,7)=[quat|trans] device or numpy arrays → refined poses (B,3,4) device float32."""
    ctx = ctx or (pose_src.context if isinstance(pose_src, DeviceArray) else Context.get(0))
    if not isinstance(pose_src, DeviceArray):
        pose_src = ctx.array(pose_src)
    if not isinstance(se3, DeviceArray):
        se3 = ctx.array(se3)
    B = pose_src.shape[0]
    out = out if out is not None else ctx.empty((B, 3, 4))
    if rot_coord.lower() not in ROT_COORD_CODE:
        raise Exception("Unknown rot_coord in R_transform: {}".format(rot_coord))
    lib.deepim_rt_transform(ctx.handle, out, None, pose_src, se3,
                            np.ascontiguousarray(T_means, np.float32), np.ascontiguousarray(T_stds, np.float32),
                            ROT_COORD_CODE[rot_coord.lower()], B)
    return out


def RT_transform(pose_src, r, t, T_means, T_stds, rot_coord="MODEL", ctx=None):
    """Single-pose call with the reference's signature; returns a (3,4) float64 numpy array."""
    r = np.squeeze(np.asarray(r))
    if r.shape[0] != 4:
        raise Exception("Unknown r shape: {}".format(r.shape)) if r.shape[0] != 3 else NotImplementedError(
            "Euler input: only ROT_TYPE=QUAT is on the device path")
    ctx = ctx or Context.get(0)
    se3 = np.concatenate([r.astype(np.float32), np.squeeze(np.asarray(t)).astype(np.float32)])[None]
    out64 = ctx.empty((1, 3, 4), dtype=np.float64)
    out = ctx.empty((1, 3, 4))
    lib.deepim_rt_transform(ctx.handle, out, out64, ctx.array(np.asarray(pose_src, np.float32)[None]), ctx.array(se3),
                            np.ascontiguousarray(T_means, np.float32), np.ascontiguousarray(T_stds, np.float32),
                            ROT_COORD_CODE[rot_coord.lower()], 1)
    return out64.asnumpy()[0]


def calc_KT(pose_src, pose_tgt, K, ctx=None):
    """K · (pose_tgt ∘ pose_src⁻¹) for a batch → (B,3,4) device float32 (input of lib/flow_c `_flow`)."""
    ctx = ctx or (pose_src.context if isinstance(pose_src, DeviceArray) else Context.get(0))
    if not isinstance(pose_src, DeviceArray):
        pose_src = ctx.array(pose_src)
    if not isinstance(pose_tgt, DeviceArray):
        pose_tgt = ctx.array(pose_tgt)
    B = pose_src.shape[0]
    out = ctx.empty((B, 3, 4))
    lib.deepim_calc_KT(ctx.handle, out, pose_src, pose_tgt, np.ascontiguousarray(K, np.float32).reshape(3, 3), B)
    return out
