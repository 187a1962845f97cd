"""Z-group kernels against the REFERENCE-GENERATED fixture (tests/golden/zoom_golden.npz: the reference's own
zoom_*.py lines run over a fake mxnet under NumPy-1.x promotion, see tests/golden/make_zoom_golden.py) — directly,
not via the oracle: zoom factors, inverse factors, crop indices and the full small-frame op outputs, all bit-exact."""
import os

import numpy as np
import pytest

from mx_deepim_amd.runtime import lib

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "zoom_golden.npz"))
H, W = 480, 640
MEANS_REV = np.ascontiguousarray(np.array([123.68, 116.779, 103.939], np.float32)[::-1])


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rect_mask(rects, H, W):
    m = np.zeros((len(rects), 1, H, W), np.float32)
    for b, (x0, x1, y0, y1) in enumerate(rects):
        if x0 >= 0:
            m[b, 0, y0:y1 + 1, x0:x1 + 1] = 1
    return m


def test_forward_factor_kernel_matches_reference_lines(ctx):
    real, rend, pose, K = G["fac_real"], G["fac_rend"], G["fac_pose"], G["fac_K"]
    n, step = len(real), 60
    got = np.zeros((n, 4), np.float32)
    for s in range(0, n, step):
        e = min(n, s + step)
        B = e - s
        mo, mr = ctx.array(rect_mask(real[s:e], H, W)), ctx.array(rect_mask(rend[s:e], H, W))
        o = [ctx.empty((B, 1, H, W)) for _ in range(3)]
        zf = ctx.empty((B, 4))
        lib.deepim_zoom_mask_forward(ctx.handle, mo, mo, mr, ctx.array(pose[s:e]), K, o[0], o[1], o[2], zf, B, H, W)
        got[s:e] = zf.asnumpy()
    np.testing.assert_array_equal(bits(got), bits(G["fac_zoom_mask_legacy"]))
    # the NumPy-2 reading of the same lines is a different fixture: the kernel must NOT match it everywhere
    assert (bits(got) != bits(G["fac_zoom_mask_np2"])).any()


def test_zoom_image_factor_matches_reference_lines(ctx):
    real, rend, pose, K = G["fac_real"][:240], G["fac_rend"][:240], G["fac_pose"][:240], G["fac_K"]
    got = np.zeros((240, 4), np.float32)
    m = MEANS_REV.reshape(1, 3, 1, 1)
    for s in range(0, 240, 40):
        e = s + 40
        io = (np.repeat(rect_mask(real[s:e], H, W), 3, 1) * np.float32(90) - m).astype(np.float32)
        ir = (np.repeat(rect_mask(rend[s:e], H, W), 3, 1) * np.float32(40) - m).astype(np.float32)
        o0, o1, zf = ctx.empty(io.shape), ctx.empty(io.shape), ctx.empty((40, 4))
        lib.deepim_zoom_image_forward(ctx.handle, ctx.array(io), ctx.array(ir), ctx.array(pose[s:e]), K, MEANS_REV, o0, o1,
                                      zf, 40, H, W)
        got[s:e] = zf.asnumpy()
    np.testing.assert_array_equal(bits(got), bits(G["fac_zoom_image_legacy"]))


def test_inverse_factor_kernel_matches_reference_lines(ctx):
    zf = G["inv_in"]
    out = ctx.empty(zf.shape)
    lib.deepim_zoom_inverse_factor(ctx.handle, ctx.array(zf), out, len(zf), H, W)
    np.testing.assert_array_equal(bits(out.asnumpy()), bits(G["inv_flow_legacy"]))
    np.testing.assert_array_equal(bits(out.asnumpy()), bits(G["inv_mask_legacy"]))


def test_crop_indices_match_materialised_grid(ctx):
    sel = G["idx_sel"]
    for zf, x0, y0 in ((G["fac_zoom_mask_legacy"][sel], G["idx_fwd_x0"], G["idx_fwd_y0"]),
                       (G["inv_flow_legacy"][sel], G["idx_inv_x0"], G["idx_inv_y0"])):
        B = len(zf)
        idx = ctx.empty((B, 2, H, W), dtype=np.int32)
        lib.deepim_zoom_indices(ctx.handle, ctx.array(zf), idx, B, H, W)
        got = idx.asnumpy()
        assert (got[:, 0] == got[:, 0, 0:1, :]).all() and (got[:, 1] == got[:, 1, :, 0:1]).all()
        np.testing.assert_array_equal(got[:, 0, 0, :], np.clip(x0, -4, W + 4))   # far-outside indices are clamped
        np.testing.assert_array_equal(got[:, 1, :, 0], np.clip(y0, -4, H + 4))


def _s(k):
    return G["small_%s_legacy" % k]


def test_every_zoom_op_matches_reference_outputs(ctx):
    """All seven Z ops through the C ABI at the 60x80 frame of the fixture (one empty rendered mask)."""
    h, w, B = 60, 80, 4
    K, pose = np.ascontiguousarray(_s("K")), ctx.array(_s("pose"))
    s4 = (B, 1, h, w)
    o = [ctx.empty(s4) for _ in range(3)]
    zf = ctx.empty((B, 4))
    lib.deepim_zoom_mask_forward(ctx.handle, ctx.array(_s("mo")), ctx.array(_s("mgt")), ctx.array(_s("depth_r")), pose, K,
                                 o[0], o[1], o[2], zf, B, h, w)
    for got, key in zip(o + [zf], ("zm0", "zm1", "zm2", "zf")):
        np.testing.assert_array_equal(bits(got.asnumpy()), bits(_s(key)), err_msg=key)
    io, ir = ctx.array(_s("io")), ctx.array(_s("ir"))
    o0, o1, zf2 = ctx.empty(_s("io").shape), ctx.empty(_s("io").shape), ctx.empty((B, 4))
    lib.deepim_zoom_image_forward(ctx.handle, io, ir, pose, K, MEANS_REV, o0, o1, zf2, B, h, w)
    for got, key in zip((o0, o1, zf2), ("zi0", "zi1", "zi_zf")):
        np.testing.assert_array_equal(bits(got.asnumpy()), bits(_s(key)), err_msg=key)
    for hl in (0, 1):
        lib.deepim_zoom_image_with_factor_forward(ctx.handle, zf, io, ir, MEANS_REV, hl, o0, o1, B, h, w)
        np.testing.assert_array_equal(bits(o0.asnumpy()), bits(_s("ziwf0_hl%d" % hl)))
        np.testing.assert_array_equal(bits(o1.asnumpy()), bits(_s("ziwf1_hl%d" % hl)))
    d0, d1 = ctx.empty(s4), ctx.empty(s4)
    lib.deepim_zoom_depth_forward(ctx.handle, zf, ctx.array(_s("dobs")), ctx.array(_s("depth_r")), d0, d1, B, h, w)
    np.testing.assert_array_equal(bits(d0.asnumpy()), bits(_s("zd0")))
    np.testing.assert_array_equal(bits(d1.asnumpy()), bits(_s("zd1")))
    f0, f1 = ctx.empty(_s("flow").shape), ctx.empty(_s("flow").shape)
    lib.deepim_zoom_flow_forward(ctx.handle, zf, ctx.array(_s("flow")), ctx.array(_s("wts")), f0, f1, 0, B, h, w)
    np.testing.assert_array_equal(bits(f0.asnumpy()), bits(_s("zflow")))
    np.testing.assert_array_equal(bits(f1.asnumpy()), bits(_s("zflow_w")))
    lib.deepim_zoom_flow_forward(ctx.handle, zf, ctx.array(_s("flow")), None, f0, None, 1, B, h, w)
    np.testing.assert_array_equal(bits(f0.asnumpy()), bits(_s("zflow_inv")))
    om, ot = ctx.empty(s4), ctx.empty((B, 3))
    for inv in (0, 1):
        lib.deepim_zoom_mask_with_factor_forward(ctx.handle, zf, ctx.array(_s("mask_in")), om, inv, B, h, w)
        np.testing.assert_array_equal(bits(om.asnumpy()), bits(_s("zmwf_inv%d" % inv)))
        lib.deepim_zoom_trans_forward(ctx.handle, zf, ctx.array(_s("trans")), ot, inv, B)
        np.testing.assert_array_equal(bits(ot.asnumpy()), bits(_s("ztrans_inv%d" % inv)))
        for zg in (0, 1):
            lib.deepim_zoom_trans_backward(ctx.handle, zf, ctx.array(_s("trans")), ot, inv, zg, B)
            np.testing.assert_array_equal(bits(ot.asnumpy()), bits(_s("ztrans_bwd_inv%d_zg%d" % (inv, zg))))
