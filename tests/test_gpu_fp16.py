"""fp16 conv path (BASELINE config 5) on the GPU vs the oracle's fp16 emulation (fp16-rounded operands and
outputs, fp32 accumulation). Tolerances: one fp16 ulp (2^-11 ≈ 4.9e-4 relative) per layer output; the pose
lands within 2e-3 of the fp32 path on the synthetic pairs — fp16 cannot meet the fp32 path's 1e-4 bar."""
import copy
import ctypes

import numpy as np
import pytest

from oracle import net as onet
from oracle import pipeline as opipe
from mx_deepim_amd.config import default_config
from mx_deepim_amd.runtime import DeviceArray, lib
from mx_deepim_amd.symbols import deepIM_flownet
from mx_deepim_amd import synthetic

pytestmark = pytest.mark.gpu
cf = ctypes.c_float
MEANS_REV = np.ascontiguousarray(synthetic.PIXEL_MEANS[::-1])


def _conv_f16(ctx, x, w, b, s, p, slope):
    B, cin, H, W = x.shape
    cout, _, k, _ = w.shape
    cpad = (cin + 7) // 8 * 8
    xh = ctx.empty((B, H, W, cpad), dtype=np.float16)
    lib.deepim_nchw_f32_to_nhwc_f16(ctx.handle, xh, ctx.array(x), B, cin, H, W, cpad)
    pk = DeviceArray(ctx, (lib.load().deepim_conv_f16_packed_size(cout, cpad, k, k) // 2,), dtype=np.float16)
    lib.deepim_conv_f16_pack_weights(ctx.handle, pk, ctx.array(w), cout, cin, cpad, k, k)
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    oh = ctx.empty((B, ho, wo, cout), dtype=np.float16)
    lib.deepim_conv2d_f16_forward(ctx.handle, oh, xh, pk, ctx.array(b), B, cpad, H, W, cout, k, k, s, p, cf(slope))
    out = ctx.empty((B, cout, ho, wo))
    lib.deepim_nhwc_f16_to_nchw_f32(ctx.handle, out, oh, B, cout, ho, wo)
    return out.asnumpy()


@pytest.mark.parametrize("case", [(2, 8, 96, 128, 64, 7, 2, 3), (1, 64, 60, 80, 128, 5, 2, 2), (2, 24, 15, 20, 256, 3, 1, 1),
                                  (2, 5, 9, 11, 68, 3, 1, 1), (1, 128, 30, 40, 512, 3, 2, 1), (3, 1024, 8, 10, 1024, 3, 1, 1),
                                  (4, 64, 120, 160, 256, 3, 1, 1)])   # last: 300 tiles → tail split of the DMA kernel
def test_conv_f16_matches_emulation(ctx, case):
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    got = _conv_f16(ctx, x, w, b, s, p, 0.1)
    ref = opipe.q16(onet.conv2d(opipe.q16(x), opipe.q16(w), b, s, p, 0.1))
    # same operands, fp32 accumulation in a different order, then one fp16 rounding: ≤ 1 fp16 ulp apart
    assert np.abs(got - ref).max() <= 2.0 ** -10 * np.maximum(1.0, np.abs(ref)).max()
    assert np.mean(got != ref) < 0.02


@pytest.mark.parametrize("case", [(3, 64, 120, 160, 256, 3, 1, 1),      # 225 tiles of 256x256, K = 18 stages
                                  (6, 64, 240, 320, 128, 5, 2, 2),      # conv2's geometry: 225 tiles of 128x512 (Cout = 128)
                                  (1, 128, 231, 233, 512, 3, 1, 1),     # two M tiles, ragged last pixel tile, odd frame
                                  (13, 128, 120, 160, 256, 5, 2, 2),    # conv3's geometry: stride 2, 5x5, 100 stages
                                  (4, 64, 120, 160, 256, 3, 1, 1)])     # 300 tiles: 256 whole + a tail split of 44
def test_conv_f16_pingpong_kernel(ctx, case):
    """conv_f16_pp_kernel (round 4: one 8-wave block per CU, the two waves of a SIMD alternating between fragment reads + LDS-DMA
    issue and the MFMAs, 4- / 5-stage ring with counted vmcnt) on grids of >= 200 tiles: against the fp16 emulation (<= 1 fp16 ulp),
    and — with K splitting off, so that both kernels add a pixel's products in the same order — BIT-IDENTICAL to the 4-wave
    kernel of round 3 (f16_dev_flags bit 16 selects it). Three repeats: a DMA / barrier ordering slip shows as rare wrong tiles."""
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    got = _conv_f16(ctx, x, w, b, s, p, 0.1)
    ref = opipe.q16(onet.conv2d(opipe.q16(x), opipe.q16(w), b, s, p, 0.1))
    assert np.abs(got - ref).max() <= 2.0 ** -10 * np.maximum(1.0, np.abs(ref)).max()
    assert np.mean(got != ref) < 0.02
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    try:
        new = [_conv_f16(ctx, x, w, b, s, p, 0.1) for _ in range(4)]
        lib.deepim_set_option(ctx.handle, b"f16_dev_flags", 16)
        old = _conv_f16(ctx, x, w, b, s, p, 0.1)
    finally:
        lib.deepim_set_option(ctx.handle, b"f16_dev_flags", 0)
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    for n_ in new:
        np.testing.assert_array_equal(n_, old)


def test_layout_round_trip(ctx):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 5, 7, 9)).astype(np.float32)
    xh = ctx.empty((2, 7, 9, 8), dtype=np.float16)
    lib.deepim_nchw_f32_to_nhwc_f16(ctx.handle, xh, ctx.array(x), 2, 5, 7, 9, 8)
    h = xh.asnumpy()
    np.testing.assert_array_equal(h[..., :5], x.astype(np.float16).transpose(0, 2, 3, 1))
    assert not h[..., 5:].any()
    xh5 = ctx.array(np.ascontiguousarray(h[..., :5]), dtype=np.float16)
    back = ctx.empty((2, 5, 7, 9))
    lib.deepim_nhwc_f16_to_nchw_f32(ctx.handle, back, xh5, 2, 5, 7, 9)
    np.testing.assert_array_equal(back.asnumpy(), x.astype(np.float16).astype(np.float32))


def test_fp16_iteration_vs_emulation_and_fp32(ctx, small_batch):
    d = small_batch
    B = d["image_observed"].shape[0]
    cfg = default_config()
    cfg.network.FP16_CONV = True
    net = deepIM_flownet().get_symbol(cfg)
    params = net.init_weights(cfg, seed=7)
    net.bind(ctx, B, params)
    data = {k: ctx.array(d[k]) for k in ("image_observed", "mask_observed")}
    data.update({k: ctx.array(d[k][0]) for k in ("image_rendered", "mask_rendered", "src_pose")})
    pose = net.refine_iteration(data).asnumpy()
    npd = {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][0], "mask_observed": d["mask_observed"],
           "mask_rendered": d["mask_rendered"][0], "src_pose": d["src_pose"][0]}
    args = (params, npd, d["K"], MEANS_REV, cfg.dataset.trans_means, cfg.dataset.trans_stds, cfg.network.ROT_COORD)
    emu = opipe.refine_iteration(*args, fp16_conv=True)
    c = net.act["conv6_1"].asnumpy()
    assert np.abs(c - emu["conv6_1"]).max() <= 2e-3 * np.abs(emu["conv6_1"]).max()
    assert np.abs(net.act["se3"].asnumpy() - emu["se3"]).max() / np.abs(emu["se3"]).max() < 1e-3
    assert np.abs(pose - emu["pose_est"]).max() / np.abs(emu["pose_est"]).max() < 1e-4
    # fused front end (default): fp16 pixel records = the fp32 net input rounded once; the two-step form (fp32 net input, then
    # conv1 converts) gives the SAME conv1 output bit for bit
    assert net._input_live_h16
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), opipe.q16(emu["net_input"]))
    h1 = net.act["flow_conv1_h"].asnumpy().copy()
    net.fp16_fused_input = False
    pose_b = net.refine_iteration(data).asnumpy()
    net.fp16_fused_input = True
    assert not net._input_live_h16
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), emu["net_input"])
    np.testing.assert_array_equal(net.act["flow_conv1_h"].asnumpy(), h1)
    np.testing.assert_array_equal(pose_b, pose)
    ref32 = opipe.refine_iteration(*args)   # how far fp16 is from the fp32 reference path (documented, not 1e-4)
    dev = np.abs(pose - ref32["pose_est"]).max() / np.abs(ref32["pose_est"]).max()
    assert dev < 2e-2, dev
    print("fp16 vs fp32 pose deviation: %.3g" % dev)


@pytest.mark.parametrize("shape", [(2, 64, 128), (1, 50, 68), (1, 480, 640)])
def test_conv1_f16_patch_kernel(ctx, shape):
    """conv1 of the fp16 path on the persistent patch kernel (NCHW fp32 in → NHWC fp16 out) vs the fp16 emulation."""
    B, H, W = shape
    rng = np.random.default_rng(H)
    x = rng.uniform(-1, 1, (B, 8, H, W)).astype(np.float32)
    w = (rng.standard_normal((64, 8, 7, 7)) / np.sqrt(8 * 49)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    ref = opipe.q16(onet.conv2d(opipe.q16(x), opipe.q16(w), b, 2, 3, 0.1))
    ho, wo = ref.shape[2:]
    pk = DeviceArray(ctx, (lib.load().deepim_conv1_x3_packed_size() // 2,), dtype=np.float16)
    lib.deepim_conv1_x3_pack_weights(ctx.handle, pk, ctx.array(w), cf(1.0))
    oh = ctx.empty((B, ho, wo, 64), dtype=np.float16)
    lib.deepim_conv1_f16_forward(ctx.handle, oh, ctx.array(x), pk, ctx.array(b), B, H, W, cf(0.1))
    out = ctx.empty((B, 64, ho, wo))
    lib.deepim_nhwc_f16_to_nchw_f32(ctx.handle, out, oh, B, 64, ho, wo)
    got = out.asnumpy()
    assert np.abs(got - ref).max() <= 2.0 ** -10 * np.maximum(1.0, np.abs(ref)).max()
    assert np.mean(got != ref) < 0.02


@pytest.mark.parametrize("shape", [(2, 64, 128), (1, 50, 68), (1, 480, 640)])
def test_conv1_f16_rgbd_patch_kernel(ctx, shape):
    """conv1 for BASELINE config 5's 10-channel RGB-D net input on the 8 + 2 channel form of the patch kernel (channels 8, 9
    as 14 pseudo taps of 4 columns x 2 channels) vs the fp16 emulation, and vs the generic fp16 conv on a 16-channel padding.
    The extra channels carry depth-like values (0 background, 0.5…1.2 m) next to image-like ones; frame borders (zero
    padding of the 7x7 window, also for the 4-column groups) are part of every shape."""
    B, H, W = shape
    rng = np.random.default_rng(H + 1)
    x = rng.uniform(-1, 1, (B, 10, H, W)).astype(np.float32)
    x[:, 8:] = np.where(rng.random((B, 2, H, W)) < 0.6, rng.uniform(0.5, 1.2, (B, 2, H, W)), 0).astype(np.float32)
    w = (rng.standard_normal((64, 10, 7, 7)) / np.sqrt(10 * 49)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    ref = opipe.q16(onet.conv2d(opipe.q16(x), opipe.q16(w), b, 2, 3, 0.1))
    ho, wo = ref.shape[2:]
    pk = DeviceArray(ctx, (lib.load().deepim_conv1_f16_c10_packed_size() // 2,), dtype=np.float16)
    lib.deepim_conv1_f16_c10_pack_weights(ctx.handle, pk, ctx.array(w))
    oh = ctx.empty((B, ho, wo, 64), dtype=np.float16)
    lib.deepim_conv1_f16_c10_forward(ctx.handle, oh, ctx.array(x), pk, ctx.array(b), B, H, W, cf(0.1))
    out = ctx.empty((B, 64, ho, wo))
    lib.deepim_nhwc_f16_to_nchw_f32(ctx.handle, out, oh, B, 64, ho, wo)
    got = out.asnumpy()
    assert np.abs(got - ref).max() <= 2.0 ** -10 * np.maximum(1.0, np.abs(ref)).max()
    assert np.mean(got != ref) < 0.02
    # the two extra channels really contribute: zeroing their weights changes the result
    w0 = w.copy()
    w0[:, 8:] = 0
    assert np.abs(opipe.q16(onet.conv2d(opipe.q16(x), opipe.q16(w0), b, 2, 3, 0.1)) - ref).max() > 0.05
    gen = _conv_f16(ctx, x, w, b, 2, 3, 0.1)           # generic kernel, Cin padded 10 → 16
    assert np.abs(got - gen).max() <= 2.0 ** -10 * np.maximum(1.0, np.abs(ref)).max()


@pytest.mark.parametrize("depth", [False, True], ids=["8ch", "rgbd_10ch"])
def test_conv1_f16_from_fp16_pixel_records(ctx, depth):
    """deepim_conv1_f16_h16_forward (input: (B,H,W,8) [+ (B,H,W,2)] fp16 records) equals deepim_conv1_f16[_c10]_forward on the
    NCHW fp32 tensor holding the same (fp16-representable) values — bit for bit, three frame sizes incl. ragged tiles."""
    for B, H, W in ((2, 64, 128), (1, 52, 68), (1, 480, 640)):
        C = 10 if depth else 8
        rng = np.random.default_rng(H + C)
        x = opipe.q16(rng.uniform(-1, 1, (B, C, H, W)).astype(np.float32))
        w = (rng.standard_normal((64, C, 7, 7)) / np.sqrt(C * 49)).astype(np.float32)
        b = rng.standard_normal(64).astype(np.float32)
        ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        if depth:
            pk = DeviceArray(ctx, (lib.load().deepim_conv1_f16_c10_packed_size() // 2,), dtype=np.float16)
            lib.deepim_conv1_f16_c10_pack_weights(ctx.handle, pk, ctx.array(w))
        else:
            pk = DeviceArray(ctx, (lib.load().deepim_conv1_x3_packed_size() // 2,), dtype=np.float16)
            lib.deepim_conv1_x3_pack_weights(ctx.handle, pk, ctx.array(w), cf(1.0))
        ref_h = ctx.empty((B, ho, wo, 64), dtype=np.float16)
        fwd = lib.deepim_conv1_f16_c10_forward if depth else lib.deepim_conv1_f16_forward
        fwd(ctx.handle, ref_h, ctx.array(x), pk, ctx.array(b), B, H, W, cf(0.1))
        main8 = ctx.array(np.ascontiguousarray(x[:, :8].transpose(0, 2, 3, 1)).astype(np.float16), dtype=np.float16)
        extra2 = ctx.array(np.ascontiguousarray(x[:, 8:].transpose(0, 2, 3, 1)).astype(np.float16), dtype=np.float16) if depth else None
        got_h = ctx.empty((B, ho, wo, 64), dtype=np.float16)
        lib.deepim_conv1_f16_h16_forward(ctx.handle, got_h, main8, extra2, pk, ctx.array(b), B, H, W, cf(0.1))
        np.testing.assert_array_equal(got_h.asnumpy(), ref_h.asnumpy(), err_msg=str((B, H, W, depth)))
