"""Golden vectors for the Z-group, produced by running the REFERENCE'S OWN operator files.

Runs only in the build container (needs /root/reference):  python tests/golden/make_zoom_golden.py
Writes tests/golden/zoom_golden.npz (committed).  TEST INFRASTRUCTURE ONLY.

/root/reference/deepim/operator_py/zoom_{mask,image,image_with_factor,depth,flow,mask_with_factor,trans}.py
are imported UNMODIFIED on top of tests/golden/fake_mxnet.py (a numpy-backed `mxnet`/`cv2` stand-in), and
their `forward`/`backward` methods are called through the CustomOp protocol, so every fixture below is an
output of the reference's own arithmetic lines:
    zoom_mask.py:47-103 / zoom_image.py:41-98   bbox -> centre -> crop -> (wx, wy, tx, ty)
    zoom_flow.py:36-44 / zoom_mask_with_factor.py:43-52   inverse factor
    zoom_*.py thresholds (>0.3, >0.2, >0.01), +means / -means, round, *wx, /wx, round(x-0.45)
Third-party (MXNet GridGenerator / BilinearSampler / round) comes from the fake's literal restatement and
is therefore still "unpinned by the reference"; the fixtures record it so the oracle's separable
formulation and the fake's materialised-grid formulation cross-check each other.

Promotion semantics: every fixture exists twice — `*_legacy` (NumPy 1.x scalar promotion, the reference's
era: MXNet 1.2 ⇒ numpy < 1.17; THE PARITY TARGET) and `*_np2` (this container's NumPy 2.2, kept as a
second reading).  See fake_mxnet.LegacyF32.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF_OPS = "/root/reference/deepim/operator_py"

import fake_mxnet as fm  # noqa: E402

f32 = np.float32
K_LM = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], f32)
MEANS = np.array([123.68, 116.779, 103.939], f32)   # cfg PIXEL_MEANS as the yaml lists them; the Prop reverses them


def import_reference_ops():
    fm.install()
    sys.path.insert(0, REF_OPS)
    import importlib
    for name in ("zoom_mask", "zoom_image", "zoom_image_with_factor", "zoom_depth", "zoom_flow",
                 "zoom_mask_with_factor", "zoom_trans"):
        importlib.import_module(name)
    assert set(fm.REGISTRY) >= {"ZoomMask", "ZoomImage", "ZoomImageWithFactor", "ZoomDepth", "ZoomFlow",
                                "ZoomMaskWithFactor", "ZoomTrans"}


def kstr(K):
    return str(np.asarray(K, f32).flatten())


def mstr(m):
    return str(np.asarray(m, f32))


def rect_mask(rects, H, W):
    """rects (B,4) int [x0,x1,y0,y1] inclusive, x0 < 0 = empty -> (B,1,H,W) f32 0/1."""
    m = np.zeros((len(rects), 1, H, W), f32)
    for b, (x0, x1, y0, y1) in enumerate(rects):
        if x0 >= 0:
            m[b, 0, y0:y1 + 1, x0:x1 + 1] = 1
    return m


def factor_cases(rng, n, H, W, K):
    """Seeded rectangles + poses: real box, rendered box (every 9th empty), centre projected near the boxes."""
    real = np.zeros((n, 4), np.int32)
    rend = np.zeros((n, 4), np.int32)
    pose = np.zeros((n, 3, 4), f32)
    for i in range(n):
        def box():
            w, h = rng.integers(2, W // 2), rng.integers(2, H // 2)
            x0, y0 = rng.integers(0, W - w), rng.integers(0, H - h)
            return [x0, x0 + w - 1, y0, y0 + h - 1]
        real[i] = box()
        rend[i] = box() if i % 9 != 4 else [-1, -1, -1, -1]
        if i % 7 == 3:                      # degenerate one-pixel / one-row boxes
            real[i, 1] = real[i, 0]
        if i % 11 == 5 and rend[i, 0] >= 0:
            rend[i, 3] = rend[i, 2]
        z = rng.uniform(0.4, 1.4)
        u, v = rng.uniform(8, W - 8), rng.uniform(8, H - 8)      # projected centre anywhere in the frame
        q = rng.standard_normal((3, 3))
        pose[i, :, :3] = np.linalg.qr(q)[0].astype(f32)
        pose[i, :, 3] = [(u - K[0, 2]) / K[0, 0] * z, (v - K[1, 2]) / K[1, 1] * z, z]
    return real, rend, pose


def run_factor_cases(mode, real, rend, pose, H, W, K, op="ZoomMask"):
    fm.set_promotion(mode)
    fm.set_sampling(False)
    n = len(real)
    out = np.zeros((n, 4), f32)
    B = 8
    for s in range(0, n, B):
        e = min(n, s + B)
        mo, mr = rect_mask(real[s:e], H, W), rect_mask(rend[s:e], H, W)
        if op == "ZoomMask":
            res, _ = fm.run_op("ZoomMask", [mo, mo, mr, pose[s:e]], [mo.shape] * 3 + [(e - s, 4)], K=kstr(K), height=H,
                               width=W)
            out[s:e] = res[3]
        else:
            mrev = MEANS[::-1].reshape(1, 3, 1, 1)
            io = (np.repeat(mo, 3, 1) * f32(90.0) - mrev).astype(f32)     # -mean outside (sum(img+mean) == 0), 90 inside
            ir = (np.repeat(mr, 3, 1) * f32(40.0) - mrev).astype(f32)
            res, _ = fm.run_op("ZoomImage", [io, ir, pose[s:e]], [io.shape] * 2 + [(e - s, 4)], K=kstr(K), height=H,
                               width=W, pixel_means=mstr(MEANS))
            out[s:e] = res[2]
    fm.captured_affines()
    fm.set_sampling(True)
    return out


def run_inverse_cases(mode, zf, H, W):
    fm.set_promotion(mode)
    fm.set_sampling(False)
    fm.captured_affines()
    n = len(zf)
    dummy2 = np.zeros((n, 2, 2, 2), f32)
    dummy1 = np.zeros((n, 1, 2, 2), f32)
    fm.run_op("ZoomFlow", [zf, dummy2], [dummy2.shape], height=H, width=W, b_inv_zoom=True)
    a_flow = np.concatenate(fm.captured_affines())
    fm.run_op("ZoomMaskWithFactor", [zf, dummy1], [dummy1.shape], height=H, width=W, b_inv_zoom=True)
    a_mask = np.concatenate(fm.captured_affines())
    fm.set_sampling(True)

    def unpack(a):   # [[wx,0,tx],[0,wy,ty]] -> (wx,wy,tx,ty)
        return np.stack([a[:, 0, 0], a[:, 1, 1], a[:, 0, 2], a[:, 1, 2]], 1).astype(f32)
    return unpack(a_flow), unpack(a_mask)


def source_indices(zf, H, W):
    """floor source indices from the fake's MATERIALISED grid (checks that it is separable)."""
    mx = sys.modules["mxnet"]
    x0 = np.zeros((len(zf), W), np.int32)
    y0 = np.zeros((len(zf), H), np.int32)
    for i, (wx, wy, tx, ty) in enumerate(zf):
        a = mx.nd.array([[wx, 0, tx], [0, wy, ty]]).reshape((1, 6))
        g = mx.nd.GridGenerator(data=a, transform_type="affine", target_shape=(H, W)).a[0]
        xr = (g[0] + f32(1)) * f32(W - 1) / f32(2)
        yr = (g[1] + f32(1)) * f32(H - 1) / f32(2)
        xi, yi = np.floor(xr).astype(np.int32), np.floor(yr).astype(np.int32)
        assert (xi == xi[0:1]).all() and (yi == yi[:, 0:1]).all(), "grid not separable"
        x0[i], y0[i] = xi[0], yi[:, 0]
    fm.captured_affines()
    return x0, y0


def small_ops(mode, rng_seed, H, W):
    """Every Z op, full outputs, at a small frame so the tensors themselves can be committed."""
    fm.set_promotion(mode)
    fm.set_sampling(True)
    rng = np.random.default_rng(rng_seed)
    B = 4
    K = K_LM.copy()
    K[:2] /= 8.0
    real, rend, pose = factor_cases(rng, B, H, W, K)
    rend[1] = [-1, -1, -1, -1]                               # one empty rendered mask: the fallback branch
    mo = rect_mask(real, H, W)
    mgt = mo.copy()
    mgt[:, :, ::7] = 0                                       # gt differs from est
    depth_r = (rect_mask(rend, H, W) * rng.uniform(0.1, 1.2, (B, 1, H, W))).astype(f32)   # depth as mask input: > 0.2 rule
    mrev = MEANS[::-1].reshape(1, 3, 1, 1)
    io = (rng.uniform(0, 255, (B, 3, H, W)).astype(f32) - mrev).astype(f32)
    ir = ((rect_mask(rend, H, W) * rng.uniform(1, 255, (B, 3, H, W))).astype(f32) - mrev).astype(f32)
    dobs = rng.uniform(0, 2, (B, 1, H, W)).astype(f32)
    flow = rng.standard_normal((B, 2, H, W)).astype(f32) * f32(5)
    wts = (rng.random((B, 2, H, W)) > 0.4).astype(f32)
    mask_in = rng.random((B, 1, H, W)).astype(f32)
    trans = rng.standard_normal((B, 3)).astype(f32)
    o = {"K": K, "real": real, "rend": rend, "pose": pose, "mo": mo, "mgt": mgt, "depth_r": depth_r, "io": io, "ir": ir,
         "dobs": dobs, "flow": flow, "wts": wts, "mask_in": mask_in, "trans": trans}
    s4 = (B, 1, H, W)
    res, _ = fm.run_op("ZoomMask", [mo, mgt, depth_r, pose], [s4, s4, s4, (B, 4)], K=kstr(K), height=H, width=W)
    o["zm0"], o["zm1"], o["zm2"], o["zf"] = res
    zf = res[3]
    res, _ = fm.run_op("ZoomImage", [io, ir, pose], [io.shape, io.shape, (B, 4)], K=kstr(K), height=H, width=W,
                       pixel_means=mstr(MEANS))
    o["zi0"], o["zi1"], o["zi_zf"] = res
    for hl in (False, True):
        res, _ = fm.run_op("ZoomImageWithFactor", [zf, io, ir], [io.shape, io.shape], height=H, width=W,
                           pixel_means=mstr(MEANS), high_light_center=hl)
        o["ziwf0_hl%d" % hl], o["ziwf1_hl%d" % hl] = res
    import contextlib
    import io as _io
    with contextlib.redirect_stdout(_io.StringIO()):          # zoom_depth.py:32 prints per sample
        res, _ = fm.run_op("ZoomDepth", [zf, dobs, depth_r], [s4, s4], height=H, width=W)
    o["zd0"], o["zd1"] = res
    res, _ = fm.run_op("ZoomFlow", [zf, flow, wts], [flow.shape, flow.shape], height=H, width=W, b_inv_zoom=False)
    o["zflow"], o["zflow_w"] = res
    res, _ = fm.run_op("ZoomFlow", [zf, flow], [flow.shape], height=H, width=W, b_inv_zoom=True)
    o["zflow_inv"] = res[0]
    for inv in (False, True):
        res, _ = fm.run_op("ZoomMaskWithFactor", [zf, mask_in], [s4], height=H, width=W, b_inv_zoom=inv)
        o["zmwf_inv%d" % inv] = res[0]
        res, opr = fm.run_op("ZoomTrans", [zf, trans], [(B, 3)], b_inv_zoom=inv, b_zoom_grad=False)
        o["ztrans_inv%d" % inv] = res[0]
        for zg in (False, True):
            prop = fm.REGISTRY["ZoomTrans"](b_inv_zoom=str(inv), b_zoom_grad=str(zg))
            opr = prop.create_operator(None, None, None)
            ig = [fm.NDArray(np.zeros((B, 4), f32)), fm.NDArray(np.zeros((B, 3), f32))]
            opr.backward(["write", "write"], [fm.NDArray(trans)], [fm.NDArray(zf), fm.NDArray(trans)], [], ig, [])
            o["ztrans_bwd_inv%d_zg%d" % (inv, zg)] = ig[1].a.copy()
    fm.captured_affines()
    return o


def full_size(mode):
    """ZoomMask + ZoomImageWithFactor at 480x640 on the repo's synthetic pairs (B = 2, seed 2333): masks
    bit-packed, images as sha256 (the inputs are regenerated by the test from mx_deepim_amd.synthetic)."""
    from mx_deepim_amd import synthetic
    fm.set_promotion(mode)
    fm.set_sampling(True)
    d = synthetic.make_batch(2, seed=2333, n_frames=1)
    H, W = 480, 640
    mo, mr, sp = d["mask_observed"], d["depth_rendered"][0], d["src_pose"][0]
    s4 = mo.shape
    res, _ = fm.run_op("ZoomMask", [mo, mo, mr, sp], [s4, s4, s4, (2, 4)], K=kstr(d["K"]), height=H, width=W)
    io, ir = d["image_observed"], d["image_rendered"][0]
    res2, _ = fm.run_op("ZoomImageWithFactor", [res[3], io, ir], [io.shape, io.shape], height=H, width=W,
                        pixel_means=mstr(synthetic.PIXEL_MEANS))
    fm.captured_affines()
    return {"zf": res[3], "zm0_bits": np.packbits(res[0].astype(np.uint8)), "zm2_bits": np.packbits(res[2].astype(np.uint8)),
            "zi0_sha": np.frombuffer(hashlib.sha256(res2[0].tobytes()).digest(), np.uint8),
            "zi1_sha": np.frombuffer(hashlib.sha256(res2[1].tobytes()).digest(), np.uint8)}


def main():
    import_reference_ops()
    H, W = 480, 640
    rng = np.random.default_rng(2333)
    out = {}
    real, rend, pose = factor_cases(rng, 1200, H, W, K_LM)
    out.update(fac_real=real, fac_rend=rend, fac_pose=pose, fac_K=K_LM)
    for mode in ("legacy", "numpy2"):
        tag = "legacy" if mode == "legacy" else "np2"
        out["fac_zoom_mask_" + tag] = run_factor_cases(mode, real, rend, pose, H, W, K_LM, "ZoomMask")
        out["fac_zoom_image_" + tag] = run_factor_cases(mode, real[:240], rend[:240], pose[:240], H, W, K_LM, "ZoomImage")
    # inverse factors: the forward factors above + random ones
    extra = np.stack([rng.uniform(0.05, 2.5, 600), np.zeros(600), rng.uniform(-1.2, 1.2, 600), rng.uniform(-1.2, 1.2, 600)],
                     1).astype(f32)
    extra[:, 1] = extra[:, 0]
    inv_in = np.concatenate([out["fac_zoom_mask_legacy"], extra]).astype(f32)
    out["inv_in"] = inv_in
    for mode in ("legacy", "numpy2"):
        tag = "legacy" if mode == "legacy" else "np2"
        out["inv_flow_" + tag], out["inv_mask_" + tag] = run_inverse_cases(mode, inv_in, H, W)
    # source ("crop") indices of the forward and inverse zooms, legacy factors
    sel = np.arange(0, 1200, 10)
    out["idx_sel"] = sel
    out["idx_fwd_x0"], out["idx_fwd_y0"] = source_indices(out["fac_zoom_mask_legacy"][sel], H, W)
    out["idx_inv_x0"], out["idx_inv_y0"] = source_indices(out["inv_flow_legacy"][sel], H, W)
    for mode in ("legacy", "numpy2"):
        tag = "legacy" if mode == "legacy" else "np2"
        for k, v in small_ops(mode, 77, 60, 80).items():
            if mode == "legacy" or k in ("zf", "zi_zf", "zflow_inv", "zmwf_inv1"):
                out["small_%s_%s" % (k, tag)] = v
        for k, v in full_size(mode).items():
            out["full_%s_%s" % (k, tag)] = v
    path = os.path.join(HERE, "zoom_golden.npz")
    np.savez_compressed(path, **out)
    d_leg_np2 = int((out["fac_zoom_mask_legacy"].view(np.uint32) != out["fac_zoom_mask_np2"].view(np.uint32)).any(1).sum())
    i_leg_np2 = int((out["inv_flow_legacy"].view(np.uint32) != out["inv_flow_np2"].view(np.uint32)).any(1).sum())
    print("wrote %s (%.1f KB): %d arrays; forward factors differing legacy vs numpy2: %d/1200; inverse: %d/%d"
          % (path, os.path.getsize(path) / 1024, len(out), d_leg_np2, i_leg_np2, len(inv_in)))


if __name__ == "__main__":
    main()
