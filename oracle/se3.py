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


def _mm(A, B, accum):
    """float32 GEMM under one of the two readings of MXNet's `batch_dot` (a BLAS sgemm whose order of accumulation is
    not specified): "seq" = the K terms of an output added one at a time in float32, left to right, unfused;
    "f64" = float64 accumulation rounded once.  Same switch as tests/golden/fake_mxnet.py:set_accum."""
    A, B = np.asarray(A, f32), np.asarray(B, f32)
    if accum == "f64":
        return (A.astype(f64) @ B.astype(f64)).astype(f32)
    out = np.zeros((A.shape[0], B.shape[1]), f32)
    for k in range(A.shape[1]):
        out = (out + (A[:, k:k + 1] * B[k:k + 1, :]).astype(f32)).astype(f32)
    return out


def _sum_last(x, accum):
    """mx.nd.sum(x, axis=-1) of float32 data under the same two readings."""
    x = np.asarray(x, f32)
    if accum == "f64":
        return x.astype(f64).sum(axis=-1).astype(f32)
    return np.add.accumulate(x, axis=-1, dtype=f32)[..., -1]


def _t3d_T_transform(T_src, T_delta, T_means, T_stds, rot_coord):
    """RT_transform.py:74-95 as transform3d.py:88-93 calls it: every operand float32 (the Prop parses T_means / T_stds
    with dtype=np.float32, transform3d.py:288-289), so the whole chain runs in float32; the float64 T_tgt it is stored
    into holds float32 values."""
    Ts, Td = np.asarray(T_src, f32), np.asarray(T_delta, f32)
    mu, sd = np.asarray(T_means, f32), np.asarray(T_stds, f32)
    assert Ts[2] != 0, "T_src: {}".format(Ts)
    d1 = ((Td * sd).astype(f32) + mu).astype(f32)
    z2 = f32(Ts[2] / np.exp(d1[2], dtype=f32))
    rc = rot_coord.lower()
    if rc in ("camera", "model"):
        return np.array([z2 * f32(d1[0] + f32(Ts[0] / Ts[2])), z2 * f32(d1[1] + f32(Ts[1] / Ts[2])), z2], f32)
    if rc == "camera_new":
        return np.array([f32(f32(Ts[2] * d1[0]) + Ts[0]), f32(f32(Ts[2] * d1[1]) + Ts[1]), z2], f32)
    raise Exception("Unknown: {}".format(rot_coord))


def transform3d_forward(points, rotation, translation, pose_src, T_means, T_stds, rot_coord="MODEL", accum="seq",
                        host_blas=False):
    """transform3dOperator.forward (transform3d.py:34-97). points (B,3,N) -> (B,3,N) float32.
    Pinned by tests/golden/ops_golden.npz (the reference file run unmodified): Rm_delta bit-exact; the output bit-exact
    with `host_blas=True`, ~1 ulp otherwise — Rm_tgt (and NAIVE's R·T_src) go through the reference's own np.dot
    (R_transform / T_transform_naive, RT_transform.py:47-60,98-102), i.e. whatever sgemm the host BLAS runs for a 3x3
    product; the default restates it as an unfused sequential float32 sum, `host_blas` calls this host's np.dot."""
    points = np.asarray(points, f32)
    B = points.shape[0]
    out = np.zeros_like(points)
    rc = rot_coord.lower()
    dot3 = (lambda A, B_: np.dot(np.asarray(A, f32), np.asarray(B_, f32))) if host_blas else _dot3_f32
    for b in range(B):
        P = np.asarray(pose_src[b], f32)
        Rd = t3d_quat2mat_forward(rotation[b])
        Rt = dot3(P[:, :3], Rd) if rc == "model" else dot3(Rd, P[:, :3])
        if rc == "naive":
            Tt = (dot3(Rd, P[:, 3:4])[:, 0] + np.asarray(translation[b], f32)).astype(f32)
        else:
            Tt = _t3d_T_transform(P[:, 3], translation[b], T_means, T_stds, rot_coord)
        out[b] = (_mm(Rt, points[b], accum) + Tt[:, None]).astype(f32)
    return out


def transform3d_backward(out_grad, points, rotation, translation, pose_src, T_means, T_stds, rot_coord="MODEL", accum="f64"):
    """transform3dOperator.backward (transform3d.py:99-151, :153-183, :214-281) -> (d_rot (B,4), d_trans (B,3)).
    NumPy-1.x promotion written out: in quat2mat_backward the products of two float32 scalars are float32, a Python
    int times a float32 scalar is float64 (`2 * x_`, `0 * D[0, 0]`), sums mixing the two are float64; the NDArray
    arithmetic of T_transform_backward is float32 throughout.  `accum` picks the reading of the third-party reductions
    (see _mm).  Bit-exact against the reference-run fixture under both readings (tests/test_oracle_ops_golden.py)."""
    out_grad, points = np.asarray(out_grad, f32), np.asarray(points, f32)
    B = points.shape[0]
    rc = rot_coord.lower()
    d_rot, d_trans = np.zeros((B, 4), f32), np.zeros((B, 3), f32)
    mu, sd = np.asarray(T_means, f32), np.asarray(T_stds, f32)
    for b in range(B):
        P = np.asarray(pose_src[b], f32)
        g = out_grad[b]
        Dt = _sum_last(g, accum)                                   # T_tgt_diff, transform3d.py:115
        Td, Tsrc = np.asarray(translation[b], f32), P[:, 3]
        if rc == "naive":
            d_trans[b] = Dt
        else:                                                       # T_transform_backward, :153-183 (float32 NDArray ops)
            d1 = ((Td * sd).astype(f32) + mu).astype(f32)
            z2 = f32(Tsrc[2] / np.exp(d1[2], dtype=f32))
            share = f32(f32(-sd[2]) * z2)
            if rc in ("camera", "model"):
                d_trans[b, 0] = f32(Dt[0] * f32(sd[0] * z2))
                d_trans[b, 1] = f32(Dt[1] * f32(sd[1] * z2))
                a0 = f32(Dt[0] * f32(share * f32(d1[0] + f32(Tsrc[0] / Tsrc[2]))))
                a1 = f32(Dt[1] * f32(share * f32(d1[1] + f32(Tsrc[1] / Tsrc[2]))))
                d_trans[b, 2] = f32(f32(a0 + a1) + f32(Dt[2] * share))
            else:
                d_trans[b, 0] = f32(Dt[0] * f32(sd[0] * Tsrc[2]))
                d_trans[b, 1] = f32(Dt[1] * f32(sd[1] * Tsrc[2]))
                d_trans[b, 2] = f32(Dt[2] * share)
        # R diff, :128-141
        Rsrc = P[:, :3]
        if rc == "naive":
            src = (_mm(Rsrc, points[b], accum) + P[:, 3:4]).astype(f32)
            D = _mm(g, src.T, accum)
        else:
            Dr = _mm(g, points[b].T, accum)
            D = _mm(Rsrc.T, Dr, accum) if rc == "model" else _mm(Dr, Rsrc.T, accum)
        # quat2mat_backward, :214-281
        q = np.asarray(rotation[b], f32)
        w, x, y, z = q
        Nq = f32(f32(f32(w * w + x * x) + y * y) + z * z)
        if not (-1e-4 < f64(Nq) - 1.0 < 1e-4):
            continue
        Ns = f32(np.sqrt(Nq))
        w_, x_, y_, z_ = (q / Ns).astype(f32)

        def p32(a, c):            # float32 scalar x float32 scalar
            return f64(f32(a * c))

        def p64(k, a, c):         # (python int x float32 scalar) x float32 scalar: float64 throughout
            return (k * f64(a)) * f64(c)

        wd = (0.0 - p32(z_, D[0, 1]) + p32(y_, D[0, 2]) + p32(z_, D[1, 0]) + 0.0 - p32(x_, D[1, 2]) - p32(y_, D[2, 0])
              + p32(x_, D[2, 1]) + 0.0)
        xd = (0.0 + p32(y_, D[0, 1]) + p32(z_, D[0, 2]) + p32(y_, D[1, 0]) - p64(2, x_, D[1, 1]) - p32(w_, D[1, 2])
              + p32(z_, D[2, 0]) + p32(w_, D[2, 1]) - p64(2, x_, D[2, 2]))
        yd = (p64(-2, y_, D[0, 0]) + p32(x_, D[0, 1]) + p32(w_, D[0, 2]) + p32(x_, D[1, 0]) + 0.0 + p32(z_, D[1, 2])
              - p32(w_, D[2, 0]) + p32(z_, D[2, 1]) - p64(2, y_, D[2, 2]))
        zd = (p64(-2, z_, D[0, 0]) - p32(w_, D[0, 1]) + p32(x_, D[0, 2]) + p32(w_, D[1, 0]) - p64(2, z_, D[1, 1])
              + p32(y_, D[1, 2]) + p32(x_, D[2, 0]) + p32(y_, D[2, 1]) + 0.0)
        wD, xD, yD, zD = wd * 2.0, xd * 2.0, yd * 2.0, zd * 2.0
        share = (f64(Ns) ** 3) * (f64(w) * wD + f64(x) * xD + f64(y) * yD + f64(z) * zD)
        d_rot[b] = [f64(Ns) * wD - f64(w) * share, f64(Ns) * xD - f64(x) * share, f64(Ns) * yD - f64(y) * share,
                    f64(Ns) * zD - f64(z) * share]
    return d_rot, d_trans
