"""Z-group parity on the GPU (C ABI) against the CPU oracle: zoom factors, source indices and
resampled tensors bit-exact."""
import ctypes

import numpy as np
import pytest

from oracle import zoom as oz
from mx_deepim_amd.runtime import lib
from mx_deepim_amd import synthetic

pytestmark = pytest.mark.gpu
MEANS_REV = np.ascontiguousarray(synthetic.PIXEL_MEANS[::-1])


def test_zoom_mask_bit_exact(ctx, small_batch):
    d = small_batch
    B, _, H, W = d["mask_observed"].shape
    mo, mr, sp = d["mask_observed"], d["depth_rendered"][0], d["src_pose"][0]  # depth as mask input exercises >0.2
    r0, r1, r2, rzf = oz.zoom_mask(mo, mo, mr, sp, d["K"])
    o = [ctx.empty((B, 1, H, W)) for _ in range(3)]
    zf = ctx.empty((B, 4))
    lib.deepim_zoom_mask_forward(ctx.handle, ctx.array(mo), ctx.array(mo), ctx.array(mr), ctx.array(sp), d["K"],
                                 o[0], o[1], o[2], zf, B, H, W)
    np.testing.assert_array_equal(zf.asnumpy(), rzf)
    for got, ref in zip(o, (r0, r1, r2)):
        np.testing.assert_array_equal(got.asnumpy(), ref)
    # crop indices bit-exact
    idx = ctx.empty((B, 2, H, W), dtype=np.int32)
    lib.deepim_zoom_indices(ctx.handle, zf, idx, B, H, W)
    np.testing.assert_array_equal(idx.asnumpy(), oz.sample_indices(rzf, H, W))
    st = ctypes.c_int(-1)
    lib.deepim_zoom_status(ctx.handle, ctypes.byref(st))
    assert st.value == 0


def test_zoom_mask_rendered_empty_fallback_and_observed_empty_flag(ctx, small_batch):
    d = small_batch
    B, _, H, W = d["mask_observed"].shape
    mr = np.zeros_like(d["mask_rendered"][0])
    r0, _, r2, rzf = oz.zoom_mask(d["mask_observed"], d["mask_observed"], mr, d["src_pose"][0], d["K"])
    o = [ctx.empty((B, 1, H, W)) for _ in range(3)]
    zf = ctx.empty((B, 4))
    lib.deepim_zoom_mask_forward(ctx.handle, ctx.array(d["mask_observed"]), ctx.array(d["mask_observed"]), ctx.array(mr),
                                 ctx.array(d["src_pose"][0]), d["K"], o[0], o[1], o[2], zf, B, H, W)
    np.testing.assert_array_equal(zf.asnumpy(), rzf)
    np.testing.assert_array_equal(o[0].asnumpy(), r0)
    # observed mask empty: the reference raises (np.min of empty) → NaN factor + status bit
    with pytest.raises(ValueError):
        oz.zoom_mask(mr, mr, mr, d["src_pose"][0], d["K"])
    lib.deepim_zoom_mask_forward(ctx.handle, ctx.array(mr), ctx.array(mr), ctx.array(mr), ctx.array(d["src_pose"][0]),
                                 d["K"], o[0], o[1], o[2], zf, B, H, W)
    st = ctypes.c_int(0)
    lib.deepim_zoom_status(ctx.handle, ctypes.byref(st))
    assert st.value & 1
    assert np.all(np.isnan(zf.asnumpy()))


def test_identity_zoom_is_identity(ctx):
    rng = np.random.default_rng(3)
    B, H, W = 2, 48, 64
    x = rng.standard_normal((B, 1, H, W)).astype(np.float32)
    zf = np.tile(np.array([1, 1, 0, 0], np.float32), (B, 1))
    o0, o1 = ctx.empty(x.shape), ctx.empty(x.shape)
    lib.deepim_zoom_depth_forward(ctx.handle, ctx.array(zf), ctx.array(x), ctx.array(x), o0, o1, B, H, W)
    got = o0.asnumpy()
    np.testing.assert_array_equal(got, oz.zoom_depth(zf, x, x)[0])
    np.testing.assert_allclose(got, x, rtol=0, atol=1e-4)  # grid lands on pixel centres up to f32 rounding of the coordinates


def test_zoom_image_ops(ctx, small_batch):
    d = small_batch
    B, _, H, W = d["image_observed"].shape
    io, ir, sp = d["image_observed"], d["image_rendered"][0], d["src_pose"][0]
    r0, r1, rzf = oz.zoom_image(io, ir, sp, d["K"], MEANS_REV)
    o0, o1, zf = ctx.empty(io.shape), ctx.empty(io.shape), ctx.empty((B, 4))
    lib.deepim_zoom_image_forward(ctx.handle, ctx.array(io), ctx.array(ir), ctx.array(sp), d["K"], MEANS_REV, o0, o1, zf,
                                  B, H, W)
    np.testing.assert_array_equal(zf.asnumpy(), rzf)
    np.testing.assert_array_equal(o0.asnumpy(), r0)
    np.testing.assert_array_equal(o1.asnumpy(), r1)
    for hl in (0, 1):
        q0, q1 = oz.zoom_image_with_factor(rzf, io, ir, MEANS_REV, bool(hl))
        lib.deepim_zoom_image_with_factor_forward(ctx.handle, zf, ctx.array(io), ctx.array(ir), MEANS_REV, hl, o0, o1, B,
                                                  H, W)
        np.testing.assert_array_equal(o0.asnumpy(), q0)
        np.testing.assert_array_equal(o1.asnumpy(), q1)


def test_zoom_flow_mask_trans(ctx, small_batch):
    d = small_batch
    rng = np.random.default_rng(5)
    B, _, H, W = d["image_observed"].shape
    _, _, _, zf_np = oz.zoom_mask(d["mask_observed"], d["mask_observed"], d["mask_rendered"][0], d["src_pose"][0], d["K"])
    zf = ctx.array(zf_np)
    flow = rng.standard_normal((B, 2, H, W)).astype(np.float32)
    wts = (rng.random((B, 2, H, W)) > 0.5).astype(np.float32)
    r0, r1 = oz.zoom_flow(zf_np, flow, wts, False)
    o0, o1 = ctx.empty(flow.shape), ctx.empty(flow.shape)
    lib.deepim_zoom_flow_forward(ctx.handle, zf, ctx.array(flow), ctx.array(wts), o0, o1, 0, B, H, W)
    np.testing.assert_array_equal(o0.asnumpy(), r0)
    np.testing.assert_array_equal(o1.asnumpy(), r1)
    (ri,) = oz.zoom_flow(zf_np, flow, None, True)
    lib.deepim_zoom_flow_forward(ctx.handle, zf, ctx.array(flow), None, o0, None, 1, B, H, W)
    np.testing.assert_array_equal(o0.asnumpy(), ri)
    m = rng.random((B, 1, H, W)).astype(np.float32)
    om = ctx.empty(m.shape)
    for inv in (0, 1):
        lib.deepim_zoom_mask_with_factor_forward(ctx.handle, zf, ctx.array(m), om, inv, B, H, W)
        np.testing.assert_array_equal(om.asnumpy(), oz.zoom_mask_with_factor(zf_np, m, bool(inv)))
    t = rng.standard_normal((B, 3)).astype(np.float32)
    ot = ctx.empty(t.shape)
    for inv in (0, 1):
        lib.deepim_zoom_trans_forward(ctx.handle, zf, ctx.array(t), ot, inv, B)
        np.testing.assert_array_equal(ot.asnumpy(), oz.zoom_trans(zf_np, t, bool(inv)))
        for zg in (0, 1):
            lib.deepim_zoom_trans_backward(ctx.handle, zf, ctx.array(t), ot, inv, zg, B)
            np.testing.assert_array_equal(ot.asnumpy(), oz.zoom_trans_backward(zf_np, t, bool(inv), bool(zg)))
    # zoom then inverse zoom of the translation is the identity (zoom_trans.py:134-154)
    lib.deepim_zoom_trans_forward(ctx.handle, zf, ctx.array(t), ot, 0, B)
    lib.deepim_zoom_trans_forward(ctx.handle, zf, ot, ot, 1, B)
    np.testing.assert_allclose(ot.asnumpy(), t, rtol=1e-6)


def test_fused_front_end_equals_separate_ops(ctx, small_batch):
    d = small_batch
    B, _, H, W = d["image_observed"].shape
    ref, rzf = oz.net_input(d["image_observed"], d["image_rendered"][0], d["mask_observed"], d["mask_rendered"][0],
                            d["src_pose"][0], d["K"], MEANS_REV, d["depth_gt_observed"], d["depth_rendered"][0])
    x, zf = ctx.empty((B, 10, H, W)), ctx.empty((B, 4))
    lib.deepim_zoom_concat_forward(ctx.handle, ctx.array(d["image_observed"]), ctx.array(d["image_rendered"][0]),
                                   ctx.array(d["mask_observed"]), ctx.array(d["mask_rendered"][0]),
                                   ctx.array(d["depth_gt_observed"]), ctx.array(d["depth_rendered"][0]),
                                   ctx.array(d["src_pose"][0]), d["K"], MEANS_REV, x, zf, B, H, W)
    np.testing.assert_array_equal(zf.asnumpy(), rzf)
    np.testing.assert_array_equal(x.asnumpy(), ref)


def test_fused_front_end_without_masks(ctx, small_batch):
    """INPUT_MASK=False test graph (deepIM_flownet.py:594-605): ZoomImage computes the factor, C = 6."""
    d = small_batch
    B, _, H, W = d["image_observed"].shape
    ref, rzf = oz.net_input(d["image_observed"], d["image_rendered"][0], None, None, d["src_pose"][0], d["K"], MEANS_REV)
    x, zf = ctx.empty((B, 6, H, W)), ctx.empty((B, 4))
    lib.deepim_zoom_concat_forward(ctx.handle, ctx.array(d["image_observed"]), ctx.array(d["image_rendered"][0]), None, None,
                                   None, None, ctx.array(d["src_pose"][0]), d["K"], MEANS_REV, x, zf, B, H, W)
    np.testing.assert_array_equal(zf.asnumpy(), rzf)
    np.testing.assert_array_equal(x.asnumpy(), ref)


def test_resample_scalar_path_unaligned_width(ctx):
    """W % 4 != 0 takes the one-pixel-per-thread kernel; same results as the oracle."""
    rng = np.random.default_rng(8)
    B, H, W = 2, 30, 42
    x = rng.standard_normal((B, 1, H, W)).astype(np.float32)
    zf = np.array([[0.5, 0.5, 0.1, -0.2], [1.7, 1.7, -0.3, 0.25]], np.float32)
    o0, o1 = ctx.empty(x.shape), ctx.empty(x.shape)
    lib.deepim_zoom_depth_forward(ctx.handle, ctx.array(zf), ctx.array(x), ctx.array(x), o0, o1, B, H, W)
    r0, _ = oz.zoom_depth(zf, x, x)
    np.testing.assert_array_equal(o0.asnumpy(), r0)
    np.testing.assert_array_equal(o1.asnumpy(), r0)


def test_div255_replacement_is_exact_for_every_float(ctx):
    """The fused front end divides by 255 with a 5-op FMA-corrected reciprocal instead of an IEEE division sequence:
    all 2^32 float bit patterns, on the device, bit for bit."""
    n = ctypes.c_ulonglong(12345)
    lib.deepim_selfcheck_div255(ctx.handle, ctypes.byref(n))
    assert n.value == 0
