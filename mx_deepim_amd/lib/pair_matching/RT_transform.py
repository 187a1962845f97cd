"""Device-side pose algebra — mirror of lib/pair_matching/RT_transform.py of the reference.

Same names and argument meaning as the reference (`RT_transform` :127-151, `calc_RT_delta` :16-44, `calc_se3` :176-187,
`calc_rt_dist_m` :162-173, `quat2mat` :383-429, `mat2quat` :432-509, `euler2mat` :240-307, `mat2euler` :310-373,
`R_transform` :47-61, `T_transform` :74-95, `se3_q2m` :190-198), each a thin call into one HIP kernel of
libdeepim_hip.so (csrc/se3.hip), which keeps the reference's float32/float64 pattern.  Single-pose calls return numpy
arrays like the reference; the `*_batch` forms take/return device arrays and are what the resident loop uses.
There is no CPU fallback: without the library / a GPU these raise.
"""
import numpy as np

from ...config import ROT_COORD_CODE
from ...runtime import Context, DeviceArray, lib

ROT_TYPE_CODE = {"quat": 0, "euler": 1, "matrix": 2}
_ROT_LEN = {0: 4, 1: 3, 2: 9}


def _ctx_of(*arrays):
    for a in arrays:
        if isinstance(a, DeviceArray):
            return a.context
    return Context.default()


def _dev(ctx, a):
    return a if isinstance(a, DeviceArray) else ctx.array(np.ascontiguousarray(a, dtype=np.float32))


def _rot_coord(rot_coord, where):
    if rot_coord.lower() not in ROT_COORD_CODE:
        raise Exception("Unknown rot_coord in {}: {}".format(where, rot_coord))
    return ROT_COORD_CODE[rot_coord.lower()]


def _f32(v):
    return np.ascontiguousarray(v, np.float32)


# ------------------------------------------------------------------------------------------------ RT_transform ----
def RT_transform_batch(pose_src, se3, T_means, T_stds, rot_coord="MODEL", out=None, ctx=None):
    """pose_src (B,3,4); se3 (B,7) = [quat|trans] or (B,6) = [euler|trans]; device or numpy → refined poses (B,3,4) device."""
    ctx = ctx or _ctx_of(pose_src, se3)
    pose_src, se3 = _dev(ctx, pose_src), _dev(ctx, se3)
    B = pose_src.shape[0]
    out = out if out is not None else ctx.empty((B, 3, 4))
    rc = _rot_coord(rot_coord, "R_transform")
    if se3.shape[1] == 7:
        lib.deepim_rt_transform(ctx.handle, out, None, pose_src, se3, _f32(T_means), _f32(T_stds), rc, B)
    elif se3.shape[1] == 6:
        lib.deepim_rt_transform_euler(ctx.handle, out, None, pose_src, se3, _f32(T_means), _f32(T_stds), rc, B)
    else:
        raise Exception("Unknown r shape: {}".format((se3.shape[1] - 3,)))
    return out


def RT_transform(pose_src, r, t, T_means, T_stds, rot_coord="MODEL", ctx=None):
    """Reference signature (RT_transform.py:127-151): r has 4 (quaternion) or 3 (Euler, static xyz) numbers; returns a
    (3,4) float64 numpy array."""
    r = np.squeeze(np.asarray(r))
    if r.shape[0] not in (3, 4):
        raise Exception("Unknown r shape: {}".format(r.shape))
    ctx = ctx or Context.default()
    rc = _rot_coord(rot_coord, "R_transform")
    se3 = np.concatenate([r.astype(np.float32), np.squeeze(np.asarray(t)).astype(np.float32)])[None]
    out64, out = ctx.empty((1, 3, 4), dtype=np.float64), ctx.empty((1, 3, 4))
    fn = lib.deepim_rt_transform if r.shape[0] == 4 else lib.deepim_rt_transform_euler
    fn(ctx.handle, out, out64, ctx.array(np.asarray(pose_src, np.float32)[None]), ctx.array(se3), _f32(T_means), _f32(T_stds),
       rc, 1)
    return out64.asnumpy()[0]


# ----------------------------------------------------------------------------------------------- calc_RT_delta ----
def calc_RT_delta_batch(pose_src, pose_tgt, T_means, T_stds, rot_coord="MODEL", rot_type="MATRIX", ctx=None):
    """(B,3,4) × 2 → (rot (B,4 | 3 | 9), trans (B,3)) device float32: the ground-truth labels of the loader
    (data_pair.py:188-195) and the batch updater (batch_updater_py_multi.py:239-246)."""
    if rot_type.lower() not in ROT_TYPE_CODE:
        raise Exception("Unknown rot_type: {}".format(rot_type))
    ctx = ctx or _ctx_of(pose_src, pose_tgt)
    pose_src, pose_tgt = _dev(ctx, pose_src), _dev(ctx, pose_tgt)
    B, code = pose_src.shape[0], ROT_TYPE_CODE[rot_type.lower()]
    rc = ROT_COORD_CODE["naive"] if rot_coord.lower() == "naive" else _rot_coord(rot_coord, "R_inv_transform")
    rot, trans = ctx.empty((B, _ROT_LEN[code])), ctx.empty((B, 3))
    lib.deepim_calc_rt_delta_ex(ctx.handle, rot, trans, pose_src, pose_tgt, _f32(T_means), _f32(T_stds), rc, code, B)
    return rot, trans


def calc_RT_delta(pose_src, pose_tgt, T_means, T_stds, rot_coord="MODEL", rot_type="MATRIX", ctx=None):
    """Reference signature (RT_transform.py:16-44) → (r, t): r is (4,), (3,) or (3,3) by rot_type."""
    rot, trans = calc_RT_delta_batch(np.asarray(pose_src, np.float32)[None], np.asarray(pose_tgt, np.float32)[None],
                                     T_means, T_stds, rot_coord, rot_type, ctx)
    r = rot.asnumpy()[0].astype(np.float64)
    return (r.reshape(3, 3) if r.size == 9 else r), trans.asnumpy()[0].astype(np.float64)


# --------------------------------------------------------------------------------------------------- the rest ----
def calc_se3_batch(pose_src, pose_tgt, ctx=None):
    ctx = ctx or _ctx_of(pose_src, pose_tgt)
    pose_src, pose_tgt = _dev(ctx, pose_src), _dev(ctx, pose_tgt)
    B = pose_src.shape[0]
    rotm, t = ctx.empty((B, 3, 3)), ctx.empty((B, 3))
    lib.deepim_calc_se3(ctx.handle, rotm, t, pose_src, pose_tgt, B)
    return rotm, t


def calc_se3(pose_src, pose_tgt, ctx=None):
    """RT_transform.py:176-187 → (rotm (3,3), t (3,)) float32, as se3_mul/se3_inverse return them."""
    rotm, t = calc_se3_batch(np.asarray(pose_src, np.float32)[None], np.asarray(pose_tgt, np.float32)[None], ctx)
    return rotm.asnumpy()[0], t.asnumpy()[0]


def calc_KT(pose_src, pose_tgt, K, ctx=None):
    """K · (pose_tgt ∘ pose_src⁻¹) for a batch → (B,3,4) device float32 (input of lib/flow_c `_flow`)."""
    ctx = ctx or _ctx_of(pose_src, pose_tgt)
    pose_src, pose_tgt = _dev(ctx, pose_src), _dev(ctx, pose_tgt)
    B = pose_src.shape[0]
    out = ctx.empty((B, 3, 4))
    lib.deepim_calc_KT(ctx.handle, out, pose_src, pose_tgt, np.ascontiguousarray(K, np.float32).reshape(3, 3), B)
    return out


def calc_rt_dist_m(pose_src, pose_tgt, ctx=None):
    """RT_transform.py:162-173 → (rd_deg, td): geodesic rotation angle in degrees (‖logm(R_srcᵀR_tgt)‖_F/√2) and ‖ΔT‖,
    through the pose-error kernel (re, te of lib/utils/pose_error.py are the same two numbers)."""
    ctx = ctx or Context.default()
    out = ctx.empty((1, 5))
    dummy = ctx.zeros((3, 1))
    lib.deepim_pose_error(ctx.handle, out, ctx.array(np.asarray(pose_src, np.float32)[None]),
                          ctx.array(np.asarray(pose_tgt, np.float32)[None]), dummy, 1, np.eye(3, dtype=np.float32), 1, 1)
    o = out.asnumpy()[0]
    return float(o[0]), float(o[1])


def _convert(op, x, in_shape, out_shape, ctx):
    ctx = ctx or Context.default()
    x = np.ascontiguousarray(np.asarray(x, np.float32).reshape((-1,) + in_shape))
    out = ctx.empty((x.shape[0],) + out_shape, dtype=np.float64)
    lib.deepim_rot_convert(ctx.handle, out, ctx.array(x), op, x.shape[0])
    return out.asnumpy()


def quat2mat(q, ctx=None):
    """RT_transform.py:383-429: (w,x,y,z) → (3,3) float64; identity when ‖q‖² < eps."""
    return _convert(0, q, (4,), (3, 3), ctx)[0]


def mat2quat(M, ctx=None):
    """RT_transform.py:432-509: (3,3) → (w,x,y,z) float64 with w >= 0."""
    return _convert(1, M, (3, 3), (4,), ctx)[0]


def euler2mat(ai, aj, ak, axes="sxyz", ctx=None):
    """RT_transform.py:240-307 for the default static-xyz axes (what RT_transform calls it with, :131)."""
    if axes != "sxyz":
        raise NotImplementedError("euler2mat: only axes='sxyz' is on the device path")
    return _convert(2, [ai, aj, ak], (3,), (3, 3), ctx)[0]


def mat2euler(mat, axes="sxyz", ctx=None):
    """RT_transform.py:310-373 for the default static-xyz axes → (ax, ay, az)."""
    if axes != "sxyz":
        raise NotImplementedError("mat2euler: only axes='sxyz' is on the device path")
    return tuple(_convert(3, np.asarray(mat)[:3, :3], (3, 3), (3,), ctx)[0])


def se3_q2m(se3_q, ctx=None):
    """RT_transform.py:190-198: [quat|trans] (7,) → (3,4) float64 [R|t] with the quaternion normalised first."""
    se3_q = np.asarray(se3_q)
    assert se3_q.size == 7
    q = se3_q[0:4].astype(np.float32)
    out = np.zeros((3, 4))
    out[:, :3] = quat2mat(q / np.linalg.norm(q), ctx)
    out[:, 3] = se3_q[4:]
    return out


def R_transform(R_src, R_delta, rot_coord="MODEL", ctx=None):
    """RT_transform.py:47-61 via the pose-update kernel (identity translation delta)."""
    _rot_coord(rot_coord, "R_transform")
    pose = np.zeros((3, 4), np.float32)
    pose[:, :3], pose[2, 3] = R_src, 1.0
    q = mat2quat(np.asarray(R_delta, np.float32), ctx)
    coord = "CAMERA" if rot_coord.lower() in ("naive", "camera_new") else rot_coord
    return RT_transform(pose, q, np.zeros(3), np.zeros(3), np.ones(3), coord, ctx)[:, :3]


def T_transform(T_src, T_delta, T_means, T_stds, rot_coord, ctx=None):
    """RT_transform.py:74-95 via the pose-update kernel (identity rotation delta)."""
    if rot_coord.lower() not in ("camera", "model", "camera_new"):
        raise Exception("Unknown: {}".format(rot_coord))
    T_src = np.asarray(T_src)
    assert T_src[2] != 0, "T_src: {}".format(T_src)
    pose = np.zeros((3, 4), np.float32)
    pose[:, :3], pose[:, 3] = np.eye(3), T_src
    return RT_transform(pose, [1, 0, 0, 0], T_delta, T_means, T_stds, rot_coord, ctx)[:, 3]
