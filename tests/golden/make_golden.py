#!/usr/bin/env python
"""Generate golden vectors from the REFERENCE's own Python, imported from /root/reference in the build
container (it cannot travel to the GPU box; the .npz files below are committed instead).

Runs: lib/pair_matching/RT_transform.py (RT_transform, calc_RT_delta, calc_se3, quat2mat, mat2quat,
T_transform, calc_rt_dist_m, euler2quat doctest value), lib/utils/projection.py (se3_mul, se3_inverse),
lib/pair_matching/flow.py (calc_flow).  The reference is not modified: RT_transform.py needs three
names that NumPy 2 removed, injected before import (np.float, np.int, np.maximum_sctype).

Inputs are stored twice — float64 (NumPy-1.x and NumPy-2 promotion agree, tight tolerance) and float32
(the reference runs under this container's NumPy 2 promotion, so 1e-6 tolerance).

    python tests/golden/make_golden.py      # rewrites tests/golden/se3_golden.npz, flow_golden.npz
"""
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    np.float = float
    np.int = int
    np.maximum_sctype = lambda t: np.float64
    sys.path.insert(0, REF)
    from lib.pair_matching import RT_transform as RT
    from lib.pair_matching import flow as FL
    from lib.utils import projection as PJ
    return RT, FL, PJ


def rand_pose(rng):
    q = rng.standard_normal(4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    t = np.array([rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.1), rng.uniform(0.6, 1.2)])
    return np.concatenate([R, t[:, None]], 1)


def main():
    RT, FL, PJ = load_reference()
    rng = np.random.default_rng(2333)
    n = 24
    out = {}
    src = np.stack([rand_pose(rng) for _ in range(n)])
    tgt = np.stack([rand_pose(rng) for _ in range(n)])
    r = rng.standard_normal((n, 4)) * 0.3 + np.array([1.0, 0, 0, 0])
    t = rng.standard_normal((n, 3)) * 0.1
    mu, sd = np.array([0.01, -0.02, 0.03]), np.array([0.9, 1.1, 1.2])
    out.update(src=src, tgt=tgt, r=r, t=t, T_means=mu, T_stds=sd)
    for dt, tag in ((np.float64, "f64"), (np.float32, "f32")):
        for coord in ("MODEL", "CAMERA", "CAMERA_NEW", "NAIVE"):
            out["RT_transform_%s_%s" % (coord, tag)] = np.stack(
                [RT.RT_transform(src[i].astype(dt), r[i].astype(dt), t[i].astype(dt), mu, sd, coord) for i in range(n)])
        for coord in ("MODEL", "CAMERA", "CAMERA_NEW"):
            rt = [RT.calc_RT_delta(src[i].astype(dt), tgt[i].astype(dt), mu, sd, coord, "QUAT") for i in range(n)]
            out["calc_RT_delta_q_%s_%s" % (coord, tag)] = np.stack([a for a, _ in rt])
            out["calc_RT_delta_t_%s_%s" % (coord, tag)] = np.stack([b for _, b in rt])
        se3 = [RT.calc_se3(src[i].astype(dt), tgt[i].astype(dt)) for i in range(n)]
        out["calc_se3_R_%s" % tag] = np.stack([a for a, _ in se3])
        out["calc_se3_t_%s" % tag] = np.stack([b for _, b in se3])
        out["se3_mul_%s" % tag] = np.stack([PJ.se3_mul(src[i].astype(dt), tgt[i].astype(dt)) for i in range(n)])
        out["se3_inverse_%s" % tag] = np.stack([PJ.se3_inverse(src[i].astype(dt)) for i in range(n)])
        out["quat2mat_%s" % tag] = np.stack([RT.quat2mat(r[i].astype(dt)) for i in range(n)])
    out["mat2quat"] = np.stack([RT.mat2quat(src[i][:, :3]) for i in range(n)])
    out["rt_dist"] = np.array([RT.calc_rt_dist_m(src[i], tgt[i]) for i in range(n)])
    out["T_transform_CAMERA"] = np.stack([RT.T_transform(src[i][:, 3], t[i], mu, sd, "CAMERA") for i in range(n)])
    # doctest known answers (RT_transform.py:403-408, :468-472, :531-533)
    out["kat_quat2mat_identity"] = RT.quat2mat([1, 0, 0, 0])
    out["kat_quat2mat_180x"] = RT.quat2mat([0, 1, 0, 0])
    out["kat_mat2quat_diag"] = RT.mat2quat(np.diag([1, -1, -1]))
    out["kat_euler2quat_ryxz_123"] = RT.euler2quat(1, 2, 3, "ryxz")
    np.savez_compressed(os.path.join(HERE, "se3_golden.npz"), **out)

    # ---- calc_flow on a small analytic depth pair ----
    H, W = 48, 64
    K = np.array([[60.0, 0, 31.5], [0, 60.0, 23.5], [0, 0, 1]], dtype=np.float32)
    ps, pt = rand_pose(rng).astype(np.float32), None
    ps[:, 3] = [0.01, -0.02, 0.8]
    pt = ps.copy()
    ang = 0.08
    Rz = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    pt[:, :3] = Rz @ ps[:, :3]
    pt[:, 3] += np.array([0.01, 0.005, 0.02], np.float32)

    def plane_depth(pose):  # fronto-parallel disc of radius 0.12 m at the object's depth
        u, v = np.meshgrid(np.arange(W), np.arange(H))
        z = pose[2, 3]
        x, y = (u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z
        return np.where((x - pose[0, 3]) ** 2 + (y - pose[1, 3]) ** 2 < 0.12 ** 2, z, 0).astype(np.float32)

    dsrc, dtgt = plane_depth(ps), plane_depth(pt)
    # make the target depth consistent with the warped source so some pixels pass the 3e-3 test
    flow, vis, _ = FL.calc_flow(dsrc, ps, pt, K, dtgt, thresh=3e-2)
    flow_s, vis_s, _ = FL.calc_flow(dsrc, ps, pt, K, dtgt, thresh=3e-2, standard_rep=True)
    np.savez_compressed(os.path.join(HERE, "flow_golden.npz"), depth_src=dsrc, depth_tgt=dtgt, pose_src=ps, pose_tgt=pt,
                        K=K, thresh=np.float64(3e-2), flow=flow, visible=vis, flow_std=flow_s, visible_std=vis_s)
    print("visible px:", int(vis.sum()), "of", H * W)


if __name__ == "__main__":
    main()


def pose_error_golden():
    """tests/golden/pose_error_golden.npz: re / te / add / adi / arp_2d of lib/utils/pose_error.py on 6 seeded
    pose pairs with 700 model points each (run: python -c "import make_golden as m; m.pose_error_golden()")."""
    from math import cos, sin
    sys.path.insert(0, REF)
    from lib.utils import pose_error as PE
    rng = np.random.default_rng(77)
    B, N = 6, 700
    K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]])
    pts = (rng.standard_normal((B, N, 3)) * 0.04).astype(np.float32)
    gt = np.stack([rand_pose(rng) for _ in range(B)]).astype(np.float32)
    est = gt.copy()
    for b in range(B):
        a = rng.normal(0, 0.1, 3)
        Rz = np.array([[cos(a[2]), -sin(a[2]), 0], [sin(a[2]), cos(a[2]), 0], [0, 0, 1]])
        Ry = np.array([[cos(a[1]), 0, sin(a[1])], [0, 1, 0], [-sin(a[1]), 0, cos(a[1])]])
        est[b, :, :3] = (Rz @ Ry @ gt[b, :, :3].astype(np.float64)).astype(np.float32)
        est[b, :, 3] += rng.normal(0, 0.01, 3).astype(np.float32)
    out = np.zeros((B, 5))
    for b in range(B):
        Re, te, Rg, tg = (est[b, :, :3].astype(np.float64), est[b, :, 3].astype(np.float64),
                          gt[b, :, :3].astype(np.float64), gt[b, :, 3].astype(np.float64))
        P = pts[b].astype(np.float64)
        out[b] = [PE.re(Re, Rg), PE.te(te, tg), PE.add(Re, te, Rg, tg, P), PE.adi(Re, te, Rg, tg, P),
                  PE.arp_2d(Re, te, Rg, tg, P, K)]
    np.savez_compressed(os.path.join(HERE, "pose_error_golden.npz"), pose_est=est, pose_gt=gt,
                        points=pts.transpose(0, 2, 1).copy(), K=K.astype(np.float32), metrics=out)


def se3_extra_golden():
    """tests/golden/se3_extra_golden.npz (round 2): the Euler / matrix branches of the reference's RT_transform.py —
    RT_transform with a 3-element r (euler2mat, :130-131), calc_RT_delta with rot_type EULER / MATRIX (:34-41),
    euler2mat / mat2euler themselves — from float32 inputs, the dtype the device entries take.
    Run: python -c "import sys; sys.path.insert(0, 'tests/golden'); import make_golden as m; m.se3_extra_golden()" """
    RT, _, _ = load_reference()

    class _Np1Array(object):
        """NumPy-2 shim no. 4 (module attribute only, the reference file is untouched): mat2euler calls
        np.array(mat, dtype=float64, copy=False) (:347), which NumPy 1.x reads as "copy only if needed"."""
        def __getattr__(self, name):
            return getattr(np, name)

        def array(self, *a, **k):
            if k.get("copy") is False:
                k["copy"] = None
            return np.array(*a, **k)
    RT.np = _Np1Array()
    rng = np.random.default_rng(4242)
    n = 24
    src = np.stack([rand_pose(rng) for _ in range(n)]).astype(np.float32)
    tgt = np.stack([rand_pose(rng) for _ in range(n)]).astype(np.float32)
    e = (rng.standard_normal((n, 3)) * 0.4).astype(np.float32)
    t = (rng.standard_normal((n, 3)) * 0.1).astype(np.float32)
    mu, sd = np.array([0.01, -0.02, 0.03]), np.array([0.9, 1.1, 1.2])
    out = dict(src=src, tgt=tgt, euler=e, t=t, T_means=mu, T_stds=sd)
    for coord in ("MODEL", "CAMERA", "CAMERA_NEW", "NAIVE"):
        out["RT_transform_euler_%s" % coord] = np.stack([RT.RT_transform(src[i], e[i], t[i], mu, sd, coord) for i in range(n)])
    for coord in ("MODEL", "CAMERA", "CAMERA_NEW", "NAIVE"):
        for rtype in ("EULER", "MATRIX", "QUAT"):
            rt = [RT.calc_RT_delta(src[i], tgt[i], mu, sd, coord, rtype) for i in range(n)]
            out["calc_RT_delta_%s_%s_r" % (rtype, coord)] = np.stack([np.asarray(a, np.float64).reshape(-1) for a, _ in rt])
            out["calc_RT_delta_%s_%s_t" % (rtype, coord)] = np.stack([np.asarray(b, np.float64) for _, b in rt])
    out["euler2mat"] = np.stack([RT.euler2mat(e[i, 0], e[i, 1], e[i, 2]) for i in range(n)])
    out["mat2euler"] = np.stack([np.array(RT.mat2euler(src[i][:, :3])) for i in range(n)])
    out["mat2quat_f32"] = np.stack([RT.mat2quat(src[i][:, :3]) for i in range(n)])
    out["quat2mat_f32"] = np.stack([RT.quat2mat(out["mat2quat_f32"][i].astype(np.float32)) for i in range(n)])
    np.savez_compressed(os.path.join(HERE, "se3_extra_golden.npz"), **out)
    print("wrote se3_extra_golden.npz:", sorted(out)[:6], "...")
