"""Edge cases through the C ABI: empty batches, minimal and ragged shapes, error returns, degenerate inputs."""
import ctypes

import numpy as np
import pytest

from oracle import flow as oflow
from oracle import net as onet
from oracle import zoom as oz
from mx_deepim_amd.runtime import Context, DeviceArray, lib

pytestmark = pytest.mark.gpu
cf = ctypes.c_float
K = np.array([[572.4114, 0, 325.2611], [0, 573.57043, 242.04899], [0, 0, 1]], np.float32)


def test_empty_batch_is_a_noop_everywhere(ctx):
    h, d = ctx.handle, ctx.empty((4,))
    lib.deepim_zoom_mask_forward(h, d, d, d, d, K, d, d, d, d, 0, 480, 640)
    lib.deepim_zoom_concat_forward(h, d, d, d, d, None, None, d, K, K[0], d, d, 0, 480, 640)
    lib.deepim_conv2d_forward(h, d, d, d, None, 0, 8, 480, 640, 64, 7, 7, 2, 3, cf(0.1), 0, 0)
    lib.deepim_deconv4x4s2_crop_forward(h, d, d, d, None, 0, 8, 8, 10, 4, 15, 20, 1, 1, cf(0.1), 0, 0)
    lib.deepim_fc_forward(h, d, d, d, None, 0, 256, 256, cf(0.1))
    lib.deepim_rt_transform(h, d, None, d, d, None, None, 1, 0)
    lib.deepim_transform3d_forward(h, d, d, d, d, d, None, None, 1, 0, 3000)
    lib.deepim_flow_forward(h, d, d, d, d, d, K, 0, 480, 640)
    lib.deepim_point_matching_loss(h, d, d, None, d, d, None, cf(0.1), 0, cf(1), cf(1), 0, 3000)
    ctx.sync()


def test_error_returns_are_raised_not_printed(ctx):
    h, d = ctx.handle, ctx.empty((16,))
    with pytest.raises(RuntimeError, match="multiple of 4"):
        lib.deepim_fc_forward(h, d, d, d, None, 1, 7, 2, cf(1.0))
    with pytest.raises(RuntimeError, match="unknown option"):
        lib.deepim_set_option(h, b"no_such_knob", 1)
    with pytest.raises(RuntimeError, match="rot_coord"):
        lib.deepim_rt_transform(h, d, None, d, d, None, None, 9, 1)
    with pytest.raises(RuntimeError, match="larger than 7"):
        lib.deepim_conv2d_forward(h, d, d, d, None, 1, 1, 16, 16, 1, 9, 9, 1, 4, cf(1.0), 0, 0)
    with pytest.raises(RuntimeError, match="2 GiB"):   # 0x80000000 is the hardware-OOB marker of the gather; convs split
        # such batches into sub-batches (test_conv_input_over_2gib_runs_as_sub_batches), the deconv entry still refuses
        lib.deepim_deconv4x4s2_crop_forward(h, d, d, d, None, 4096, 1024, 16, 20, 2, 30, 38, 1, 1, cf(1.0), 0, 0)
    with pytest.raises(RuntimeError, match="crop exceeds"):
        lib.deepim_deconv4x4s2_crop_forward(h, d, d, d, None, 1, 2, 4, 4, 2, 10, 10, 1, 1, cf(1.0), 0, 0)
    with pytest.raises(RuntimeError):
        Context(99)


def _conv(ctx, x, w, b, s, p, slope=1.0):
    B, cin, H, W = x.shape
    cout, _, kh, kw = w.shape
    pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cout, cin, kh, kw) // 4,))
    lib.deepim_conv_pack_weights(ctx.handle, pk, ctx.array(w), cout, cin, kh, kw)
    out = ctx.empty((B, cout, (H + 2 * p - kh) // s + 1, (W + 2 * p - kw) // s + 1))
    lib.deepim_conv2d_forward(ctx.handle, out, ctx.array(x), pk, None if b is None else ctx.array(b), B, cin, H, W, cout,
                              kh, kw, s, p, cf(slope), 0, 0)
    return out.asnumpy()


@pytest.mark.parametrize("case", [(1, 1, 1, 1, 1, 1, 1, 0), (1, 1, 5, 3, 1, 3, 1, 1), (2, 3, 7, 7, 5, 7, 2, 3),
                                  (1, 2, 4, 130, 3, 1, 1, 0), (3, 5, 6, 6, 65, 3, 2, 1), (1, 17, 9, 9, 129, 5, 1, 2)])
def test_conv_minimal_and_ragged_shapes(ctx, case):
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    got = _conv(ctx, x, w, b, s, p, 0.1)
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    np.testing.assert_array_equal(got, onet.conv2d(x, w, b, s, p, 0.1))


def test_conv_propagates_nan_and_inf_only_where_they_belong(ctx):
    """Zero padding must be a true zero (hardware OOB), not 0·x of a neighbouring value."""
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    x = np.ones((1, 2, 6, 6), np.float32)
    x[0, 0, 0, 0] = np.inf
    x[0, 1, 5, 5] = np.nan
    w = np.ones((64, 2, 3, 3), np.float32)
    got = _conv(ctx, x, w, None, 1, 1)
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    ref = onet.conv2d(x, w, None, 1, 1, 1.0)
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.array_equal(np.isinf(got), np.isinf(ref))
    fin = np.isfinite(ref)
    np.testing.assert_array_equal(got[fin], ref[fin])
    assert fin[0, 0, 3, 3] and not fin[0, 0, 0, 0]


def test_zoom_object_touching_borders_and_huge_factor(ctx):
    B, H, W = 3, 480, 640
    m_obs = np.zeros((B, 1, H, W), np.float32)
    m_ren = np.zeros((B, 1, H, W), np.float32)
    m_obs[0, 0, :40, :50] = 1          # top-left corner
    m_ren[0, 0, :30, :60] = 1
    m_obs[1, 0, 400:, 600:] = 1        # bottom-right corner
    m_ren[1, 0, 420:, 590:] = 1
    m_obs[2, 0] = 1                    # whole frame → zoom-out factor > 1 (samples outside the image)
    m_ren[2, 0, 200:260, 300:380] = 1
    pose = np.tile(np.array([[1, 0, 0, 0.0], [0, 1, 0, 0.0], [0, 0, 1, 1.0]], np.float32), (B, 1, 1))
    pose[0, :, 3] = [-0.5, -0.38, 1.0]
    pose[1, :, 3] = [0.5, 0.38, 1.0]
    r0, _, r2, rzf = oz.zoom_mask(m_obs, m_obs, m_ren, pose, K)
    o = [ctx.empty((B, 1, H, W)) for _ in range(3)]
    zf = ctx.empty((B, 4))
    lib.deepim_zoom_mask_forward(ctx.handle, ctx.array(m_obs), ctx.array(m_obs), ctx.array(m_ren), ctx.array(pose), K,
                                 o[0], o[1], o[2], zf, B, H, W)
    np.testing.assert_array_equal(zf.asnumpy(), rzf)
    assert rzf[2, 0] > 1.0
    np.testing.assert_array_equal(o[0].asnumpy(), r0)
    np.testing.assert_array_equal(o[2].asnumpy(), r2)
    idx = ctx.empty((B, 2, H, W), dtype=np.int32)
    lib.deepim_zoom_indices(ctx.handle, zf, idx, B, H, W)
    np.testing.assert_array_equal(idx.asnumpy(), oz.sample_indices(rzf, H, W))


def test_flow_all_background_and_behind_camera(ctx):
    B, H, W = 2, 32, 48
    src = np.zeros((B, 1, H, W), np.float32)
    src[1] = 0.8
    tgt = np.full((B, 1, H, W), 0.8, np.float32)
    KT = np.tile(np.array([[60, 0, 24, 0], [0, 60, 16, 0], [0, 0, 1, 0]], np.float32), (B, 1, 1))
    KT[1, 2, 3] = -2.0  # projected depth negative: w/z flips sign, the bounds test must reject it
    Kinv = np.linalg.inv(np.array([[60, 0, 24], [0, 60, 16], [0, 0, 1]], np.float32)).astype(np.float32)
    flow, valid = ctx.empty((B, 2, H, W)), ctx.empty((B, 1, H, W))
    lib.deepim_flow_forward(ctx.handle, flow, valid, ctx.array(src), ctx.array(tgt), ctx.array(KT),
                            np.ascontiguousarray(Kinv), B, H, W)
    rf, rv = oflow.gpu_flow(src, tgt, KT, Kinv)
    np.testing.assert_array_equal(valid.asnumpy(), rv)
    np.testing.assert_array_equal(flow.asnumpy(), rf)
    assert not valid.asnumpy()[0].any()


def test_two_contexts_on_one_device_are_independent():
    a, b = Context(0), Context(0)
    x = np.arange(12, dtype=np.float32)
    da, db = a.array(x), b.array(x * 2)
    np.testing.assert_array_equal(da.asnumpy() * 2, db.asnumpy())
    del db                              # arrays of a context must go before the context does
    lib.deepim_destroy(b.handle)
    np.testing.assert_array_equal(da.asnumpy(), x)
