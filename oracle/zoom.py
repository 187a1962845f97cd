"""Z-group oracle: MXNet GridGenerator(affine) + BilinearSampler and the Zoom* CustomOps.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Z0 is third-party (MXNet 1.2 src/operator/grid_generator-inl.h, bilinear_sampler.cc — not
vendored under /root/reference): restated from the published operator code; PARITY
UNPINNED by reference tests, cross-checked against torch grid_sample in tests/.
"""
import numpy as np

f32 = np.float32
f64 = np.float64


def roundf(x):
    """C roundf / mx.nd.round: half away from zero (NOT numpy's half-to-even)."""
    x = np.asarray(x, dtype=f32)
    t = np.trunc(x)
    frac = np.abs(x - t)          # exact in floating point
    return np.where(frac >= f32(0.5), t + np.sign(x), t).astype(f32)


def _axis_taps(scale, trans, n):
    """1-D part of GridGenerator + BilinearSampler coordinate math for an axis of n pixels.

    x_d = -1 + i * f32(2/(n-1));  x_s = scale*x_d + trans;  x_r = (x_s + 1)*(n-1)/2 (all f32);
    returns int32 floor index, f32 weight of that tap, validity of tap 0 / tap 1.
    """
    step = f32(2.0 / (n - 1))
    i = np.arange(n, dtype=f32)
    xd = f32(-1.0) + i * step
    xs = f32(scale) * xd + f32(trans)
    xr = (xs + f32(1.0)) * f32(n - 1) / f32(2.0)
    finite = ~np.isnan(xr)
    xc = np.clip(np.where(finite, xr, f32(-4.0)), f32(-4.0), f32(n + 4.0)).astype(f32)
    xf = np.floor(xc).astype(f32)
    x0 = xf.astype(np.int32)
    w0 = (f32(1.0) - (xr - xf)).astype(f32)
    in0 = finite & (x0 >= 0) & (x0 <= n - 1)
    in1 = finite & (x0 + 1 >= 0) & (x0 + 1 <= n - 1)
    return x0, w0, in0, in1


def sample_indices(zoom_factor, H, W):
    """(B,4) factors -> int32 (B,2,H,W) [x0, y0] = floor source indices of every output pixel."""
    zoom_factor = np.asarray(zoom_factor, dtype=f32)
    B = zoom_factor.shape[0]
    out = np.zeros((B, 2, H, W), np.int32)
    for b in range(B):
        wx, wy, tx, ty = zoom_factor[b]
        x0, _, _, _ = _axis_taps(wx, tx, W)
        y0, _, _, _ = _axis_taps(wy, ty, H)
        out[b, 0] = x0[None, :]
        out[b, 1] = y0[:, None]
    return out


def bilinear_sample(img, wx, wy, tx, ty):
    """img (C,H,W) f32 -> (C,H,W) f32; zero padding per tap; float/double mix of bilinear_sampler.cc."""
    img = np.asarray(img, dtype=f32)
    C, H, W = img.shape
    x0, wx0, xin0, xin1 = _axis_taps(wx, tx, W)
    y0, wy0, yin0, yin1 = _axis_taps(wy, ty, H)
    xc0, xc1 = np.clip(x0, 0, W - 1), np.clip(x0 + 1, 0, W - 1)
    yc0, yc1 = np.clip(y0, 0, H - 1), np.clip(y0 + 1, 0, H - 1)

    def tap(yc, xc, yin, xin):
        v = img[:, yc[:, None], xc[None, :]]
        return np.where((yin[:, None] & xin[None, :])[None], v, f32(0.0)).astype(f32)

    tl, tr = tap(yc0, xc0, yin0, xin0), tap(yc0, xc1, yin0, xin1)
    bl, br = tap(yc1, xc0, yin1, xin0), tap(yc1, xc1, yin1, xin1)
    wy0_ = wy0[None, :, None]
    wx0_ = wx0[None, None, :]
    t1 = ((tl * wy0_).astype(f32) * wx0_).astype(f32)
    t2 = (tr * wy0_).astype(f32).astype(f64) * (1.0 - wx0_.astype(f64))
    t3 = bl.astype(f64) * (1.0 - wy0_.astype(f64)) * wx0_.astype(f64)
    t4 = br.astype(f64) * (1.0 - wy0_.astype(f64)) * (1.0 - wx0_.astype(f64))
    return (((t1.astype(f64) + t2) + t3) + t4).astype(f32)


def _bbox(valid):
    x_any = np.max(valid, axis=0)
    y_any = np.max(valid, axis=1)
    nz_x = np.nonzero(x_any)[0]
    nz_y = np.nonzero(y_any)[0]
    return nz_x, nz_y


def zoom_factor_from_valid(valid_real, valid_rendered, src_pose, K, H, W, promotion="legacy"):
    """zoom_mask.py:47-103 / zoom_image.py:41-98 for one sample -> f32 (wx, wy, tx, ty).

    Scalar promotion (pinned by tests/golden/zoom_golden.npz, i.e. by the reference's own lines run under both
    rules): the projected centre cx = c[0]/c[2] is a float32 scalar.  promotion="legacy" (NumPy 1.x, the
    reference's era, THE PARITY TARGET): `cx / self.width * 2 - 1` promotes float32-scalar-op-Python-int to
    float64, so tx, ty are a float64 chain rounded ONCE when stored into the float32 zoom_factor array.
    promotion="numpy2": the same expression stays float32 (three roundings).  Box distances are float64 under
    both (float32 scalar op int64 scalar)."""
    K = np.asarray(K, dtype=f32).reshape(3, 3)
    t = np.asarray(src_pose, dtype=f32)[:, 3]
    nz_x, nz_y = _bbox(valid_real)
    if len(nz_x) == 0 or len(nz_y) == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    rsx, rex, rsy, rey = f64(nz_x.min()), f64(nz_x.max()), f64(nz_y.min()), f64(nz_y.max())
    real_cx, real_cy = (rsx + rex) * 0.5, (rsy + rey) * 0.5
    nz_x, nz_y = _bbox(valid_rendered)
    # np.dot(K, t) in float32, sequential, no FMA
    c = [f32(f32(f32(K[i, 0] * t[0]) + f32(K[i, 1] * t[1])) + f32(K[i, 2] * t[2])) for i in range(3)]
    cx, cy = f32(c[0] / c[2]), f32(c[1] / c[2])
    if len(nz_x) == 0 or len(nz_y) == 0:
        osx, oex, osy, oey = rsx, rex, rsy, rey
        zcx, zcy = real_cx, real_cy
        tx = f32(zcx / W * 2 - 1)          # float64 chain, cast at the store
        ty = f32(zcy / H * 2 - 1)
    else:
        osx, oex, osy, oey = f64(nz_x.min()), f64(nz_x.max()), f64(nz_y.min()), f64(nz_y.max())
        zcx, zcy = f64(cx), f64(cy)
        if promotion == "legacy":
            tx = f32(f64(cx) / W * 2 - 1)                       # float64 chain, one rounding at the store
            ty = f32(f64(cy) / H * 2 - 1)
        else:
            tx = f32(f32(f32(cx / f32(W)) * f32(2)) - f32(1))   # NumPy 2: float32 scalar op python int -> float32
            ty = f32(f32(f32(cy / f32(H)) * f32(2)) - f32(1))
    left = max(zcx - osx, zcx - rsx)
    right = max(oex - zcx, rex - zcx)
    up = max(zcy - osy, zcy - rsy)
    down = max(rey - zcy, oey - zcy)
    crop = np.max([0.75 * right, 0.75 * left, up, down]) * 1.4 * 2
    wx = f32(crop / H)
    return np.array([wx, wx, tx, ty], dtype=f32)


def zoom_mask(mask_observed, mask_gt_observed, mask_rendered, src_pose, K):
    """ZoomMaskOperator.forward (zoom_mask.py:29-112). Inputs (B,1,H,W), src_pose (B,3,4)."""
    B, _, H, W = mask_observed.shape
    valid_real = np.sum(np.asarray(mask_gt_observed, f32), axis=1) > 0.3
    ren = np.array(mask_rendered, dtype=f32, copy=True)
    gt = ren > 0.2
    le = ren <= 0.2
    ren[gt] = 1
    ren[le] = 0
    valid_ren = np.sum(ren, axis=1) > 0.3
    zf = np.zeros((B, 4), f32)
    outs = [np.zeros((B, 1, H, W), f32) for _ in range(3)]
    for b in range(B):
        zf[b] = zoom_factor_from_valid(valid_real[b], valid_ren[b], src_pose[b], K, H, W)
        wx, wy, tx, ty = zf[b]
        outs[0][b] = roundf(bilinear_sample(mask_observed[b], wx, wy, tx, ty))
        outs[1][b] = roundf(bilinear_sample(mask_gt_observed[b], wx, wy, tx, ty))
        outs[2][b] = roundf(bilinear_sample(ren[b], wx, wy, tx, ty))
    return outs[0], outs[1], outs[2], zf


def _means(pixel_means):
    return np.asarray(pixel_means, dtype=f32).reshape(1, 3, 1, 1)


def zoom_image(image_observed, image_rendered, src_pose, K, pixel_means):
    """ZoomImageOperator.forward (zoom_image.py:26-107). pixel_means already in tensor channel order."""
    B, _, H, W = image_observed.shape
    m = _means(pixel_means)
    real = (np.asarray(image_observed, f32) + m).astype(f32)
    ren = (np.asarray(image_rendered, f32) + m).astype(f32)

    def valid(x):  # np.sum(axis=1) float32, sequential
        return ((x[:, 0] + x[:, 1]).astype(f32) + x[:, 2]).astype(f32) > 0.01

    vr, vn = valid(real), valid(ren)
    zf = np.zeros((B, 4), f32)
    o0, o1 = np.zeros_like(real), np.zeros_like(ren)
    for b in range(B):
        zf[b] = zoom_factor_from_valid(vr[b], vn[b], src_pose[b], K, H, W)
        wx, wy, tx, ty = zf[b]
        o0[b] = bilinear_sample(real[b], wx, wy, tx, ty) - m[0]
        o1[b] = bilinear_sample(ren[b], wx, wy, tx, ty) - m[0]
    return o0, o1, zf


def zoom_image_with_factor(zoom_factor, image_observed, image_rendered, pixel_means, high_light_center=False):
    """ZoomImageWithFactorOperator.forward (zoom_image_with_factor.py:31-65)."""
    B, _, H, W = image_observed.shape
    m = _means(pixel_means)
    real = (np.asarray(image_observed, f32) + m).astype(f32)
    ren = (np.asarray(image_rendered, f32) + m).astype(f32)
    o0, o1 = np.zeros_like(real), np.zeros_like(ren)
    cm = np.zeros((3, H, W), f32)
    r = 5
    cm[0, int(np.floor(H / 2.0 - r)):int(np.ceil(H / 2.0 + r)), int(np.floor(W / 2.0 - r)):int(np.ceil(W / 2.0 + r))] = 255.0
    for b in range(B):
        wx, wy, tx, ty = np.asarray(zoom_factor, f32)[b]
        a = bilinear_sample(real[b], wx, wy, tx, ty)
        c = bilinear_sample(ren[b], wx, wy, tx, ty)
        if high_light_center:
            c = np.maximum(c, cm)
        o0[b] = a - m[0]
        o1[b] = c - m[0]
    return o0, o1


def zoom_depth(zoom_factor, depth_observed, depth_rendered):
    """ZoomDepthOperator.forward (zoom_depth.py:24-44)."""
    o0, o1 = np.zeros_like(depth_observed, dtype=f32), np.zeros_like(depth_rendered, dtype=f32)
    for b in range(depth_observed.shape[0]):
        wx, wy, tx, ty = np.asarray(zoom_factor, f32)[b]
        o0[b] = bilinear_sample(depth_observed[b], wx, wy, tx, ty)
        o1[b] = bilinear_sample(depth_rendered[b], wx, wy, tx, ty)
    return o0, o1


def inverse_factor(zf, H, W, promotion="legacy"):
    """zoom_flow.py:36-44 / zoom_mask_with_factor.py:43-52 -> f32 (wx, wy, tx, ty).

    The four inputs are float32 scalars unpacked from `asnumpy()`.  promotion="legacy" (NumPy 1.x, THE PARITY
    TARGET): every line mixes them with Python ints/floats, so the whole computation is float64 and is rounded
    once when `mx.nd.array([[wx,0,tx],[0,wy,ty]])` makes the float32 affine matrix.  promotion="numpy2": every
    line stays float32.  Both are pinned by tests/golden/zoom_golden.npz."""
    wx_in, wy_in, tx_in, ty_in = [f32(v) for v in zf]
    if promotion == "legacy":
        wx_in, wy_in, tx_in, ty_in = f64(wx_in), f64(wy_in), f64(tx_in), f64(ty_in)
        with np.errstate(divide="ignore", invalid="ignore"):
            wx = 1 / wx_in
            wy = 1 / wy_in
            crop_w = wx_in * W
            crop_h = wy_in * H
            cx = tx_in * 0.5 * W + 0.5 * W
            cy = ty_in * 0.5 * H + 0.5 * H
            tx = (W * 0.5 - cx) / crop_w * 2
            ty = (H * 0.5 - cy) / crop_h * 2
        return f32(wx), f32(wy), f32(tx), f32(ty)
    Wf, Hf, half, two = f32(W), f32(H), f32(0.5), f32(2)
    with np.errstate(divide="ignore", invalid="ignore"):
        wx = f32(f32(1) / wx_in)
        wy = f32(f32(1) / wy_in)
        crop_w = f32(wx_in * Wf)
        crop_h = f32(wy_in * Hf)
        cx = f32(f32(f32(tx_in * half) * Wf) + f32(0.5 * W))
        cy = f32(f32(f32(ty_in * half) * Hf) + f32(0.5 * H))
        tx = f32(f32(f32(f32(W * 0.5) - cx) / crop_w) * two)
        ty = f32(f32(f32(f32(H * 0.5) - cy) / crop_h) * two)
    return wx, wy, tx, ty


def zoom_flow(zoom_factor, flow, flow_weights=None, b_inv_zoom=False):
    """ZoomFlowOperator.forward (zoom_flow.py:28-71)."""
    B, _, H, W = flow.shape
    zf = np.asarray(zoom_factor, f32)
    out = np.zeros_like(flow, dtype=f32)
    outw = None if b_inv_zoom else np.zeros_like(flow_weights, dtype=f32)
    for b in range(B):
        assert zf[b, 0] == zf[b, 1], "wx and wy should be equal"
        a = inverse_factor(zf[b], H, W) if b_inv_zoom else tuple(zf[b])
        s = bilinear_sample(flow[b], *a)
        out[b] = (s * zf[b, 0]).astype(f32) if b_inv_zoom else (s / zf[b, 0]).astype(f32)
        if not b_inv_zoom:
            outw[b] = roundf(bilinear_sample(flow_weights[b], *a) - f32(0.45))
    return (out,) if b_inv_zoom else (out, outw)


def zoom_mask_with_factor(zoom_factor, mask, b_inv_zoom=False):
    """ZoomMaskWithFactorOperator.forward (zoom_mask_with_factor.py:29-64)."""
    B, _, H, W = mask.shape
    zf = np.asarray(zoom_factor, f32)
    m = np.array(mask, dtype=f32, copy=True)
    gt, le = m > 0.2, m <= 0.2
    m[gt] = 1
    m[le] = 0
    out = np.zeros_like(m)
    for b in range(B):
        a = inverse_factor(zf[b], H, W) if b_inv_zoom else tuple(zf[b])
        out[b] = roundf(bilinear_sample(m[b], *a))
    return out


def zoom_trans(zoom_factor, trans_delta, b_inv_zoom=False):
    """ZoomTransOperator.forward (zoom_trans.py:22-46): wy is read from column 0 as well."""
    zf = np.asarray(zoom_factor, f32)
    td = np.asarray(trans_delta, f32)
    out = td.copy()
    wx = zf[:, 0]
    if b_inv_zoom:
        out[:, 0] = td[:, 0] * wx
        out[:, 1] = td[:, 1] * wx
    else:
        out[:, 0] = td[:, 0] / wx
        out[:, 1] = td[:, 1] / wx
    return out


def zoom_trans_backward(zoom_factor, out_grad, b_inv_zoom=False, b_zoom_grad=False):
    """ZoomTransOperator.backward (zoom_trans.py:48-74)."""
    if not b_zoom_grad:
        return np.asarray(out_grad, f32).copy()
    return zoom_trans(zoom_factor, out_grad, b_inv_zoom)


def net_input(image_observed, image_rendered, mask_observed, mask_rendered, src_pose, K, pixel_means,
              depth_observed=None, depth_rendered=None, mask_gt_observed=None):
    """Front end of the test graph (deepIM_flownet.py:563-622 + :33-62): ZoomMask (gt ≡ observed)
    → ZoomImageWithFactor [→ ZoomDepth] → /255 → Concat."""
    if mask_observed is None:   # INPUT_MASK=False: ZoomImage computes the factor (deepIM_flownet.py:594-605)
        zio, zir, zf = zoom_image(image_observed, image_rendered, src_pose, K, pixel_means)
    else:
        # test graph: mask_gt_observed is mask_observed (deepIM_flownet.py:564); training graph: the real gt mask (:392-412)
        zmo, _, zmr, zf = zoom_mask(mask_observed, mask_observed if mask_gt_observed is None else mask_gt_observed,
                                    mask_rendered, src_pose, K)
        zio, zir = zoom_image_with_factor(zf, image_observed, image_rendered, pixel_means)
    parts = [(zio / f32(255.0)).astype(f32), (zir / f32(255.0)).astype(f32)]
    if depth_observed is not None:
        zdo, zdr = zoom_depth(zf, depth_observed, depth_rendered)
        parts += [(zdo / f32(255.0)).astype(f32), (zdr / f32(255.0)).astype(f32)]
    if mask_observed is not None:
        parts += [zmo, zmr]
    return np.concatenate(parts, axis=1), zf
