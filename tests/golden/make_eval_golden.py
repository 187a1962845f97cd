"""Golden for the (n°, n cm) / ADD(-S) / arp-2D accumulation of lib/dataset/LM6D_REFINE.py, produced by the REFERENCE'S
OWN evaluate_pose / evaluate_pose_add / evaluate_pose_arp_2d (LM6D_REFINE.py:278-370, 372-512, 514-674).

Runs only in the build container (needs /root/reference):  python tests/golden/make_eval_golden.py
Writes tests/golden/eval_golden.npz (committed).  TEST INFRASTRUCTURE ONLY.

The reference class is imported unmodified; its methods only touch self.classes / num_classes / _points / _diameters,
so the object is made with object.__new__ and those four attributes (no dataset on disk).  Stand-ins injected before
import: `lib.utils.logger` (records every logger.info line — the tables leave the reference only as formatted log
lines), `cv2` (never called on this path), `scipy.integrate.simps` (renamed `simpson` in SciPy >= 1.14), and the
NumPy-2 name shims RT_transform.py needs.  The per-pose metrics the reference computes on the way (calc_rt_dist_m, re,
add, adi, arp_2d) are recorded too, so the accumulation can be checked on the CPU with exactly the reference's inputs.
"""
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

CLASSES = ["ape", "eggbox", "glue", "lamp"]            # eggbox: symmetric + the 180° z flip; glue: ADI; lamp: no poses
NUM_ITER = 4
K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], np.float32)


def rot(axis, deg):
    a = np.asarray(axis, np.float64)
    a /= np.linalg.norm(a)
    t = np.deg2rad(deg)
    Kx = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(t) * Kx + (1 - np.cos(t)) * Kx @ Kx


def make_case(rng):
    pts = {c: (rng.standard_normal((400, 3)) * np.array([0.03, 0.025, 0.02])).astype(np.float32) for c in CLASSES}
    diam = {"ape": 0.1021, "eggbox": 0.1646, "glue": 0.1758, "lamp": 0.2852}
    n_per = {"ape": 14, "eggbox": 12, "glue": 10, "lamp": 0}
    gt, est = [], []
    for c in CLASSES:
        g, e = [], [[] for _ in range(NUM_ITER)]
        for j in range(n_per[c]):
            R = rot(rng.standard_normal(3), rng.uniform(0, 180))
            t = np.array([rng.uniform(-0.15, 0.15), rng.uniform(-0.1, 0.1), rng.uniform(0.6, 1.2)])
            g.append(np.concatenate([R, t[:, None]], 1).astype(np.float32))
            ang0, tr0 = rng.uniform(4, 25), rng.uniform(0.01, 0.08)
            ax, td = rng.standard_normal(3), rng.standard_normal(3)
            td /= np.linalg.norm(td)
            for it in range(NUM_ITER):                    # refinement: the error shrinks from iteration to iteration
                f = 0.45 ** it
                Re = rot(ax, ang0 * f) @ R
                if c == "eggbox" and j % 3 == 0:          # estimates that found the eggbox turned by half a revolution
                    Re = Re @ np.diag([-1.0, -1.0, 1.0])
                e[it].append(np.concatenate([Re, (t + td * tr0 * f)[:, None]], 1).astype(np.float32))
        gt.append([g])
        est.append(e)
    return pts, diam, est, gt


def main():
    np.float = float
    np.int = int
    np.maximum_sctype = lambda t: np.float64
    import scipy.integrate
    if not hasattr(scipy.integrate, "simps"):
        scipy.integrate.simps = scipy.integrate.simpson
    import fake_mxnet as fm
    fm.install()                                          # provides the empty `cv2`
    lines = []
    fake_logger = types.ModuleType("lib.utils.logger")
    fake_logger.info = lambda msg, *a: lines.append(str(msg))
    sys.modules["lib.utils.logger"] = fake_logger
    sys.path.insert(0, REF)
    from lib.dataset.LM6D_REFINE import LM6D_REFINE
    from lib.pair_matching.RT_transform import calc_rt_dist_m
    from lib.utils import pose_error as PE
    from lib.utils.projection import se3_mul

    rng = np.random.default_rng(2333)
    pts, diam, est, gt = make_case(rng)
    ds = object.__new__(LM6D_REFINE)
    ds.classes, ds.num_classes, ds._points, ds._diameters = CLASSES, len(CLASSES), pts, diam
    cfg = types.SimpleNamespace(TEST=types.SimpleNamespace(test_iter=NUM_ITER), dataset=types.SimpleNamespace(INTRINSIC_MATRIX=K))

    out = {"classes": np.array(CLASSES), "num_iter": np.array(NUM_ITER), "K": K}
    for c in CLASSES:
        out["points_" + c] = pts[c]
        out["diameter_" + c] = np.array(diam[c])
    for ci, c in enumerate(CLASSES):
        out["gt_" + c] = np.array(gt[ci][0], np.float32).reshape(-1, 3, 4)
        out["est_" + c] = np.array(est[ci], np.float32).reshape(NUM_ITER, -1, 3, 4)
    # the per-pose metrics the three functions compute internally, recorded with the reference's own helpers
    RT_z = np.array([[-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 1, 0]])
    for ci, c in enumerate(CLASSES):
        n = len(gt[ci][0])
        m = np.zeros((NUM_ITER, n, 6))    # rd, td (after the eggbox rule), re, add-or-adi, arp_2d (after the rule), rd raw
        for it in range(NUM_ITER):
            for j in range(n):
                E, Gt = est[ci][it][j], gt[ci][0][j]
                rd, td = calc_rt_dist_m(E, Gt)
                m[it, j, 5] = rd
                if c == "eggbox" and rd > 90:
                    rd, td = calc_rt_dist_m(se3_mul(E, RT_z), Gt)
                m[it, j, 0], m[it, j, 1] = rd, td
                m[it, j, 2] = PE.re(E[:3, :3], Gt[:3, :3])
                fn = PE.adi if c in ("eggbox", "glue", "bowl", "cup") else PE.add
                m[it, j, 3] = fn(E[:3, :3], E[:, 3], Gt[:3, :3], Gt[:, 3], pts[c])
                E2 = se3_mul(E, RT_z) if (c == "eggbox" and m[it, j, 2] > 90) else E
                m[it, j, 4] = PE.arp_2d(E2[:3, :3], E2[:, 3], Gt[:3, :3], Gt[:, 3], pts[c], K)
        out["metrics_" + c] = m

    def section(fn, *a):
        del lines[:]
        fn(*a)
        return np.array(list(lines))

    tmp = tempfile.mkdtemp()
    out["log_evaluate_pose"] = section(ds.evaluate_pose, cfg, est, gt)
    out["log_evaluate_pose_add"] = section(ds.evaluate_pose_add, cfg, est, gt, tmp)
    add_file = [f for f in os.listdir(tmp) if f.endswith("_xys.pkl")]
    out["add_pkl_name"] = np.array(add_file[0])
    with open(os.path.join(tmp, add_file[0]), "rb") as f:
        pd = pickle.load(f)
    for c, curves in pd.items():
        out["add_curve_" + c] = np.stack([np.asarray(y, np.float64) for _, y in curves])
        out["add_curve_x"] = np.asarray(curves[0][0])
    os.remove(os.path.join(tmp, add_file[0]))
    out["log_evaluate_pose_arp_2d"] = section(ds.evaluate_pose_arp_2d, cfg, est, gt, tmp)
    arp_file = [f for f in os.listdir(tmp) if f.endswith(".pkl")]
    out["arp_pkl_name"] = np.array(arp_file[0])
    with open(os.path.join(tmp, arp_file[0]), "rb") as f:
        pd = pickle.load(f)
    for c, curves in pd.items():
        out["arp_curve_" + c] = np.stack([np.asarray(y, np.float64) for _, y in curves])
        out["arp_curve_x"] = np.asarray(curves[0][0])
    path = os.path.join(HERE, "eval_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f KB): %d log lines (pose) / %d (add) / %d (arp_2d)" % (
        path, os.path.getsize(path) / 1024, len(out["log_evaluate_pose"]), len(out["log_evaluate_pose_add"]),
        len(out["log_evaluate_pose_arp_2d"])))
    for l in out["log_evaluate_pose"][:12]:
        print("  |", l)
    for l in out["log_evaluate_pose_add"][:8]:
        print("  |", l)


if __name__ == "__main__":
    main()
