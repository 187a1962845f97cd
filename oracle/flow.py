"""F-group oracle: depth warp / ground-truth flow.  TEST INFRASTRUCTURE ONLY.

F1 follows lib/flow_c/gpu_flow_kernel.cu:32-69 (the only CUDA kernel; cannot run here —
PARITY UNPINNED by reference tests, the restatement is line-by-line float32);
F2 follows lib/pair_matching/flow.py:12-63 and is pinned against that module imported here;
F3 follows deepim/operator_py/flow_updater.py:42-102.
"""
import numpy as np

from .zoom import roundf
from .se3 import calc_se3

f32 = np.float32
f64 = np.float64


def gpu_flow(depth_src, depth_tgt, KT, Kinv):
    """`flow_kernel` semantics. depth (n,1,h,w), KT (n,3,4), Kinv (3,3) f32 -> flow (n,2,h,w), valid (n,1,h,w)."""
    depth_src, depth_tgt = np.asarray(depth_src, f32), np.asarray(depth_tgt, f32)
    KT, Kinv = np.asarray(KT, f32), np.asarray(Kinv, f32).reshape(9)
    n, _, H, W = depth_src.shape
    flow = np.zeros((n, 2, H, W), f32)
    valid = np.zeros((n, 1, H, W), f32)
    hh, ww = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    wf, hf = ww.astype(f32), hh.astype(f32)
    for b in range(n):
        d = depth_src[b, 0]
        k = KT[b].reshape(12)
        x = ((wf * Kinv[0] + hf * Kinv[1]).astype(f32) + Kinv[2]).astype(f32) * d
        y = ((wf * Kinv[3] + hf * Kinv[4]).astype(f32) + Kinv[5]).astype(f32) * d
        z = d
        xp = (((x * k[0] + y * k[1]).astype(f32) + z * k[2]).astype(f32) + k[3]).astype(f32)
        yp = (((x * k[4] + y * k[5]).astype(f32) + z * k[6]).astype(f32) + k[7]).astype(f32)
        zs = (((x * k[8] + y * k[9]).astype(f32) + z * k[10]).astype(f32) + k[11]).astype(f32)
        zp = (zs.astype(f64) + 1e-15).astype(f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            wp = (xp / zp).astype(f32)
            hp = (yp / zp).astype(f32)
        ok = d.astype(f64) > 1e-3
        inb = (wp >= 0) & (wp <= f32(W - 1)) & (hp >= 0) & (hp <= f32(H - 1))
        ok &= inb
        wi = np.where(ok, roundf(np.where(ok, wp, 0)), 0).astype(np.int64)
        hi = np.where(ok, roundf(np.where(ok, hp, 0)), 0).astype(np.int64)
        dt = depth_tgt[b, 0][hi, wi]
        ok &= np.abs(zp - dt).astype(f32).astype(f64) < 3e-3
        flow[b, 0] = np.where(ok, hp - hf, 0)
        flow[b, 1] = np.where(ok, wp - wf, 0)
        valid[b, 0] = ok
    return flow, valid


def calc_KT(pose_src, pose_tgt, K):
    """batch_updater_py_multi.py:255-259: K · (pose_tgt ∘ pose_src⁻¹), float32."""
    K = np.asarray(K, f32).reshape(3, 3)
    out = np.zeros((len(pose_src), 3, 4), f32)
    for b in range(len(pose_src)):
        R, t = calc_se3(np.asarray(pose_src[b], f32), np.asarray(pose_tgt[b], f32))
        m = np.concatenate([R, t.reshape(3, 1)], axis=1).astype(f32)
        out[b] = ((K[:, 0:1] * m[0:1] + K[:, 1:2] * m[1:2]).astype(f32) + K[:, 2:3] * m[2:3]).astype(f32)
    return out


def calc_flow(depth_src, KT, Kinv, depth_tgt, thresh=3e-3, standard_rep=False):
    """flow.py:12-63 given transform = K·se3 (f32) and Kinv = inv(K) (f32). -> flow (H,W,2), visible (H,W)."""
    depth_src, depth_tgt = np.asarray(depth_src), np.asarray(depth_tgt)
    H, W = depth_src.shape[:2]
    x, y = np.meshgrid(np.arange(W), np.arange(H))
    x2d = np.stack((x, y, np.ones((H, W), dtype=f32)), axis=2).reshape(W * H, 3)
    R = np.asarray(Kinv, f32).astype(f64) @ x2d.transpose().astype(f64)
    X = np.tile(depth_src.reshape(1, W * H), (3, 1)).astype(f64) * R
    Xp = np.asarray(KT, f32).astype(f64) @ np.append(X, np.ones([1, X.shape[1]]), axis=0)
    pz = Xp[2] + 1e-15
    pw, ph = Xp[0] / pz, Xp[1] / pz
    visible = np.zeros(H * W)
    vp = np.where(depth_src.flatten() != 0)[0]
    pwr = np.round(pw[vp]).astype(int)
    phr = np.round(ph[vp]).astype(int)
    pwc = np.minimum(np.maximum(pwr, 0), W - 1)
    phc = np.minimum(np.maximum(phr, 0), H - 1)
    within = (pwr >= 0) & (pwr < W) & (phr >= 0) & (phr < H)
    dt = depth_tgt[phc, pwc]
    within &= np.abs(dt - pz[vp]) < thresh
    visible[vp[within & (np.abs(dt) > 1e-10)]] = 1
    visible = visible.reshape(H, W)
    w_ori, h_ori = np.meshgrid(np.linspace(0, W - 1, W), np.linspace(0, H - 1, H))
    if standard_rep:
        flow = np.dstack([pw.reshape(H, W) - w_ori, ph.reshape(H, W) - h_ori])
    else:
        flow = np.dstack([ph.reshape(H, W) - h_ori, pw.reshape(H, W) - w_ori])
    flow[np.dstack([visible, visible]) != 1] = 0
    return flow, visible


def flow_updater(depth_src, depth_tgt, pose_src, pose_tgt, K, thresh=3e-3, wh_rep=False):
    """flowUpdaterOperator.forward (flow_updater.py:42-102) -> flow (B,2,H,W), flow_weights (B,2,H,W)."""
    depth_src, depth_tgt = np.asarray(depth_src, f32), np.asarray(depth_tgt, f32)
    B, _, H, W = depth_src.shape
    K = np.asarray(K, f32).reshape(3, 3)
    Kinv = np.linalg.inv(K.astype(f64))
    x, y = np.meshgrid(np.arange(W), np.arange(H))
    R = (Kinv[:, 0:1] * x.reshape(1, -1) + Kinv[:, 1:2] * y.reshape(1, -1) + Kinv[:, 2:3]).astype(f32)
    T = calc_KT(pose_src, pose_tgt, K)
    flow = np.zeros((B, 2, H, W), f32)
    wts = np.zeros((B, 2, H, W), f32)
    for b in range(B):
        d = depth_src[b, 0].reshape(-1)
        X, Y, Z = d * R[0], d * R[1], d * R[2]
        t = T[b].reshape(12)

        def row(i):
            return (((t[i] * X + t[i + 1] * Y).astype(f32) + t[i + 2] * Z).astype(f32) + t[i + 3]).astype(f32)

        wp, hp, zp = row(0), row(4), (row(8) + f32(1e-15)).astype(f32)
        with np.errstate(divide="ignore", invalid="ignore"):
            pw = np.minimum(np.maximum(roundf(wp / zp), 0), f32(W - 1))
            ph = np.minimum(np.maximum(roundf(hp / zp), 0), f32(H - 1))
        pw = np.nan_to_num(pw).astype(np.int64)
        ph = np.nan_to_num(ph).astype(np.int64)
        ok = (d > f32(1e-10)) & (np.abs(depth_tgt[b, 0][ph, pw] - zp) < f32(thresh))
        wd = np.where(ok, pw - x.reshape(-1), 0).astype(f32).reshape(H, W)
        hd = np.where(ok, ph - y.reshape(-1), 0).astype(f32).reshape(H, W)
        flow[b, 0], flow[b, 1] = (wd, hd) if wh_rep else (hd, wd)
        wts[b, 0] = wts[b, 1] = ok.reshape(H, W)
    return flow, wts


def mask_box(mask):
    """data_pair.py:94-105 / image.py:363-372: rectangle [y_start:y_end, x_start:x_end] over the non-zeros of a (H,W) mask
    (end indices exclusive, as the numpy slices of the reference are); zeros for an empty mask."""
    mask = np.asarray(mask)
    out = np.zeros(mask.shape, f32)
    nz_x = np.nonzero(np.max(mask, 0))[0]
    nz_y = np.nonzero(np.max(mask, 1))[0]
    if len(nz_x) and len(nz_y):
        out[np.min(nz_y):np.max(nz_y), np.min(nz_x):np.max(nz_x)] = 1.0
    return out
