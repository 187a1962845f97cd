"""S-group oracle: SE(3) algebra, RT_transform, Transform3D forward/backward.

TEST INFRASTRUCTURE ONLY.  Follows lib/pair_matching/RT_transform.py, lib/utils/projection.py and
deepim/operator_py/transform3d.py with NumPy-1.x scalar promotion written out explicitly.
Pinned against the reference module imported in the build container (tests/golden/se3_*.npz).
"""
import math

import numpy as np

f32 = np.float32
f64 = np.float64
_FLOAT_EPS = np.finfo(np.float64).eps


def se3_inverse(RT):
    """projection.py:12-23 — result float32."""
    RT = np.asarray(RT)
    R = RT[0:3, 0:3]
    T = RT[0:3, 3].reshape((3, 1))
    out = np.zeros((3, 4), dtype=f32)
    out[0:3, 0:3] = R.transpose()
    out[0:3, 3] = -1 * np.dot(R.transpose(), T).reshape((3))
    return out


def se3_mul(RT1, RT2):
    """projection.py:26-43 — result float32."""
    RT1, RT2 = np.asarray(RT1), np.asarray(RT2)
    out = np.zeros((3, 4), dtype=f32)
    out[0:3, 0:3] = np.dot(RT1[0:3, 0:3], RT2[0:3, 0:3])
    out[0:3, 3] = (np.dot(RT1[0:3, 0:3], RT2[0:3, 3].reshape((3, 1))) + RT1[0:3, 3].reshape((3, 1))).reshape((3))
    return out


def calc_se3(pose_src, pose_tgt):
    """RT_transform.py:176-187."""
    m = se3_mul(pose_tgt, se3_inverse(pose_src))
    return m[:, :3], m[:, 3].reshape((3))


def quat2mat(q):
    """RT_transform.py:383-429. float32 scalars promote to float64 at `2.0 / Nq` (NumPy 1.x)."""
    w, x, y, z = q
    Nq = w * w + x * x + y * y + z * z
    if Nq < _FLOAT_EPS:
        return np.eye(3)
    s = 2.0 / f64(Nq)
    w, x, y, z = f64(w), f64(x), f64(y), f64(z)
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array(
        [[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]]
    )


def mat2quat(M):
    """RT_transform.py:432-509 (Bar-Itzhack, eigh of the 4x4 K matrix, w >= 0)."""
    Qxx, Qyx, Qzx, Qxy, Qyy, Qzy, Qxz, Qyz, Qzz = np.asarray(M, dtype=f64).flat
    K = np.array(
        [
            [Qxx - Qyy - Qzz, 0, 0, 0],
            [Qyx + Qxy, Qyy - Qxx - Qzz, 0, 0],
            [Qzx + Qxz, Qzy + Qyz, Qzz - Qxx - Qyy, 0],
            [Qyz - Qzy, Qzx - Qxz, Qxy - Qyx, Qxx + Qyy + Qzz],
        ]
    ) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    if q[0] < 0:
        q *= -1
    return q


def R_transform(R_src, R_delta, rot_coord="MODEL"):
    """RT_transform.py:47-61."""
    rc = rot_coord.lower()
    if rc == "model":
        return np.dot(R_src, R_delta)
    if rc in ("camera", "naive", "camera_new"):
        return np.dot(R_delta, R_src)
    raise Exception("Unknown rot_coord in R_transform: {}".format(rot_coord))


def R_inv_transform(R_src, R_tgt, rot_coord):
    """RT_transform.py:64-71."""
    rc = rot_coord.lower()
    if rc == "model":
        return np.dot(R_src.transpose(), R_tgt)
    if rc in ("camera", "camera_new"):
        return np.dot(R_tgt, R_src.transpose())
    raise Exception("Unknown rot_coord in R_inv_transform: {}".format(rot_coord))


def T_transform(T_src, T_delta, T_means, T_stds, rot_coord):
    """RT_transform.py:74-95. T_src/T_delta float32 arrays, means/stds float64."""
    T_src = np.asarray(T_src)
    assert T_src[2] != 0, "T_src: {}".format(T_src)
    d1 = np.asarray(T_delta).astype(f64) * np.asarray(T_stds, f64) + np.asarray(T_means, f64)
    T = np.zeros((3,))
    z2 = f64(T_src[2]) / np.exp(d1[2])
    T[2] = z2
    rc = rot_coord.lower()
    if rc in ("camera", "model"):
        T[0] = z2 * (d1[0] + f64(T_src[0] / T_src[2]))   # the quotient keeps T_src's dtype
        T[1] = z2 * (d1[1] + f64(T_src[1] / T_src[2]))
    elif rc == "camera_new":
        T[0] = f64(T_src[2]) * d1[0] + f64(T_src[0])
        T[1] = f64(T_src[2]) * d1[1] + f64(T_src[1])
    else:
        raise Exception("Unknown: {}".format(rot_coord))
    return T


def T_inv_transform(T_src, T_tgt, T_means, T_stds, rot_coord):
    """RT_transform.py:105-124."""
    T_src, T_tgt = np.asarray(T_src), np.asarray(T_tgt)
    d = np.zeros((3,))
    rc = rot_coord.lower()
    if rc == "camera_new":
        d[0] = (T_tgt[0] - T_src[0]) / T_src[2]
        d[1] = (T_tgt[1] - T_src[1]) / T_src[2]
    elif rc in ("camera", "model"):
        d[0] = T_tgt[0] / T_tgt[2] - T_src[0] / T_src[2]
        d[1] = T_tgt[1] / T_tgt[2] - T_src[1] / T_src[2]
    else:
        raise Exception("Unknown: {}".format(rot_coord))
    d[2] = np.log(T_src[2] / T_tgt[2])
    return (d - np.asarray(T_means, f64)) / np.asarray(T_stds, f64)


def RT_transform(pose_src, r, t, T_means, T_stds, rot_coord="MODEL"):
    """RT_transform.py:127-151 (quaternion input). -> (3,4) float64."""
    pose_src = np.asarray(pose_src)
    r = np.squeeze(np.asarray(r))
    assert r.shape[0] == 4
    if r.dtype == f32:
        nrm = f32(np.sqrt(f32(f32(f32(r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]) + r[3] * r[3])))
    else:
        nrm = np.linalg.norm(r)
    quat = r / nrm
    Rm_delta = quat2mat(quat)
    t_delta = np.squeeze(np.asarray(t))
    if rot_coord.lower() == "naive":
        se3_mx = np.zeros((3, 4))
        se3_mx[:, :3] = Rm_delta
        se3_mx[:, 3] = t_delta
        return se3_mul(se3_mx, pose_src)
    pose_est = np.zeros((3, 4))
    pose_est[:3, :3] = R_transform(pose_src[:3, :3], Rm_delta, rot_coord)
    pose_est[:3, 3] = T_transform(pose_src[:, 3], t_delta, T_means, T_stds, rot_coord)
    return pose_est


def calc_RT_delta(pose_src, pose_tgt, T_means, T_stds, rot_coord="MODEL", rot_type="MATRIX"):
    """RT_transform.py:16-44."""
    if rot_coord.lower() == "naive":
        m = se3_mul(pose_tgt, se3_inverse(pose_src))
        Rm_delta, T_delta = m[:, :3], m[:, 3].reshape((3))
    else:
        Rm_delta = R_inv_transform(pose_src[:3, :3], pose_tgt[:3, :3], rot_coord)
        T_delta = T_inv_transform(pose_src[:, 3], pose_tgt[:, 3], T_means, T_stds, rot_coord)
    if rot_type.lower() == "quat":
        r = mat2quat(Rm_delta)
    elif rot_type.lower() == "matrix":
        r = Rm_delta
    else:
        raise Exception("Unknown rot_type: {}".format(rot_type))
    return r, np.squeeze(T_delta)


def calc_rt_dist_m(pose_src, pose_tgt):
    """RT_transform.py:162-173 (rotation geodesic via the trace instead of scipy logm; same value)."""
    R = np.dot(np.transpose(pose_src[:, :3]).astype(f64), np.asarray(pose_tgt[:, :3], f64))
    c = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    rd_deg = math.acos(c) / math.pi * 180
    td = np.linalg.norm(np.asarray(pose_tgt[:, 3], f64) - np.asarray(pose_src[:, 3], f64))
    return rd_deg, td


# ------------------------------------------------------------------ Transform3D ----
def t3d_quat2mat_forward(q):
    """transform3d.py:185-212 — identity unless |Nq-1| < 1e-2; float32 result."""
    w, x, y, z = [f32(v) for v in q]
    Nq = f32(f32(f32(w * w + x * x) + y * y) + z * z)
    if not (-1e-2 < f64(f32(Nq - f32(1))) < 1e-2):
        return np.eye(3, dtype=f32)
    s = 2.0 / f64(Nq)
    w, x, y, z = f64(w), f64(x), f64(y), f64(z)
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    return np.array(
        [[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]]
    ).astype(f32)


def _dot3_f32(A, B):
    """3x3·3xN float32 with sequential unfused accumulation."""
    A, B = np.asarray(A, f32), np.asarray(B, f32)
    return ((A[:, 0:1] * B[0:1] + A[:, 1:2] * B[1:2]).astype(f32) + A[:, 2:3] * B[2:3]).astype(f32)


def transform3d_forward(points, rotation, translation, pose_src, T_means, T_stds, rot_coord="MODEL"):
    """transform3dOperator.forward (transform3d.py:34-97). points (B,3,N) -> (B,3,N) float32."""
    points = np.asarray(points, f32)
    B = points.shape[0]
    out = np.zeros_like(points)
    rc = rot_coord.lower()
    for b in range(B):
        P = np.asarray(pose_src[b], f32)
        Rd = t3d_quat2mat_forward(rotation[b])
        Rt = _dot3_f32(P[:, :3], Rd) if rc == "model" else _dot3_f32(Rd, P[:, :3])
        if rc == "naive":
            Tt = (_dot3_f32(Rd, P[:, 3:4])[:, 0] + np.asarray(translation[b], f32)).astype(f32)
        else:
            Tt = T_transform(P[:, 3], np.asarray(translation[b], f32), T_means, T_stds, rot_coord).astype(f32)
        out[b] = (_dot3_f32(Rt, points[b]) + Tt[:, None]).astype(f32)
    return out


def transform3d_backward(out_grad, points, rotation, translation, pose_src, T_means, T_stds, rot_coord="MODEL"):
    """transform3dOperator.backward (transform3d.py:99-151, :153-183, :214-281) -> (d_rot (B,4), d_trans (B,3))."""
    out_grad, points = np.asarray(out_grad, f32), np.asarray(points, f32)
    B = points.shape[0]
    rc = rot_coord.lower()
    d_rot, d_trans = np.zeros((B, 4), f32), np.zeros((B, 3), f32)
    mu, sd = np.asarray(T_means, f32), np.asarray(T_stds, f32)
    for b in range(B):
        P = np.asarray(pose_src[b], f32)
        g = out_grad[b].astype(f64)  # accumulate in float64; the device reduces in float32 (tolerance test)
        Dt = g.sum(axis=1).astype(f32)
        src = points[b]
        if rc == "naive":
            src = (_dot3_f32(P[:, :3], points[b]) + P[:, 3:4]).astype(f32)
        Dr = (g @ src.astype(f64).T).astype(f32)
        Td, Tsrc = np.asarray(translation[b], f32), P[:, 3]
        if rc == "naive":
            d_trans[b] = Dt
        else:
            d1 = (Td * sd + mu).astype(f32)
            z2 = f32(Tsrc[2] / np.exp(d1[2], dtype=f32))
            if rc in ("camera", "model"):
                share = f32(-sd[2] * z2)
                d_trans[b, 0] = Dt[0] * f32(sd[0] * z2)
                d_trans[b, 1] = Dt[1] * f32(sd[1] * z2)
                d_trans[b, 2] = (Dt[0] * (share * (d1[0] + Tsrc[0] / Tsrc[2])) + Dt[1] * (share * (d1[1] + Tsrc[1] / Tsrc[2]))
                                 + Dt[2] * (-sd[2] * z2))
            else:
                d_trans[b, 0] = Dt[0] * f32(sd[0] * Tsrc[2])
                d_trans[b, 1] = Dt[1] * f32(sd[1] * Tsrc[2])
                d_trans[b, 2] = Dt[2] * (-sd[2] * z2)
        if rc == "model":
            D = _dot3_f32(P[:, :3].T, Dr)
        elif rc == "naive":
            D = Dr
        else:
            D = _dot3_f32(Dr, P[:, :3].T)
        q = np.asarray(rotation[b], f32)
        w, x, y, z = q
        Nq = f32(f32(f32(w * w + x * x) + y * y) + z * z)
        if not (-1e-4 < f64(f32(Nq - f32(1))) < 1e-4):
            continue
        Ns = f32(np.sqrt(Nq))
        w_, x_, y_, z_ = (q / Ns).astype(f32)
        wd = (0 * D[0, 0] - z_ * D[0, 1] + y_ * D[0, 2] + z_ * D[1, 0] + 0 * D[1, 1] - x_ * D[1, 2] - y_ * D[2, 0]
              + x_ * D[2, 1] + 0 * D[2, 2])
        xd = (0 * D[0, 0] + y_ * D[0, 1] + z_ * D[0, 2] + y_ * D[1, 0] - 2 * x_ * D[1, 1] - w_ * D[1, 2] + z_ * D[2, 0]
              + w_ * D[2, 1] - 2 * x_ * D[2, 2])
        yd = (-2 * y_ * D[0, 0] + x_ * D[0, 1] + w_ * D[0, 2] + x_ * D[1, 0] + 0 * D[1, 1] + z_ * D[1, 2] - w_ * D[2, 0]
              + z_ * D[2, 1] - 2 * y_ * D[2, 2])
        zd = (-2 * z_ * D[0, 0] - w_ * D[0, 1] + x_ * D[0, 2] + w_ * D[1, 0] - 2 * z_ * D[1, 1] + y_ * D[1, 2] + x_ * D[2, 0]
              + y_ * D[2, 1] + 0 * D[2, 2])
        wD, xD, yD, zD = f64(f32(wd)) * 2.0, f64(f32(xd)) * 2.0, f64(f32(yd)) * 2.0, f64(f32(zd)) * 2.0
        share = f64(f32(Ns * Ns * Ns)) * (f64(w) * wD + f64(x) * xD + f64(y) * yD + f64(z) * zD)
        d_rot[b] = [f64(Ns) * wD - f64(w) * share, f64(Ns) * xD - f64(x) * share, f64(Ns) * yD - f64(y) * share,
                    f64(Ns) * zD - f64(z) * share]
    return d_rot, d_trans
