"""N-group parity on the GPU (C ABI) against the C oracle: conv / deconv bit-exact (both are
k-ordered fp32 fmaf chains), FC within 1e-5 relative."""
import ctypes

import numpy as np
import pytest

from oracle import net as onet
from mx_deepim_amd.runtime import DeviceArray, lib

pytestmark = pytest.mark.gpu
cf = ctypes.c_float


def _pack_conv(ctx, w):
    cout, cin, kh, kw = w.shape
    pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cout, cin, kh, kw) // 4,))
    lib.deepim_conv_pack_weights(ctx.handle, pk, ctx.array(w), cout, cin, kh, kw)
    return pk


def _run_conv(ctx, x, w, b, s, p, slope):
    B, cin, H, W = x.shape
    cout, _, kh, kw = w.shape
    ho, wo = (H + 2 * p - kh) // s + 1, (W + 2 * p - kw) // s + 1
    out = ctx.empty((B, cout, ho, wo))
    lib.deepim_conv2d_forward(ctx.handle, out, ctx.array(x), _pack_conv(ctx, w), None if b is None else ctx.array(b), B,
                              cin, H, W, cout, kh, kw, s, p, cf(slope), 0, 0)
    return out.asnumpy()


# (B, Cin, H, W, Cout, k, s, p): every kernel geometry of the encoder at reduced spatial size, the
# small-Cout heads, ragged sizes, each tile config (128x128, 64x128, 128x64, 64x64)
CASES = [
    (2, 8, 96, 128, 64, 7, 2, 3),     # flow_conv1 geometry
    (1, 64, 60, 80, 128, 5, 2, 2),    # conv2
    (2, 16, 33, 47, 256, 5, 2, 2),    # conv3, ragged spatial
    (3, 24, 15, 20, 256, 3, 1, 1),    # conv3_1
    (2, 32, 30, 40, 512, 3, 2, 1),    # conv4
    (2, 1024, 8, 10, 2, 3, 1, 1),     # Convolution1 (Cout = 2)
    (1, 770, 30, 40, 1, 3, 1, 1),     # mask_conv3 (Cout = 1, Cin not a multiple of 16)
    (1, 3, 9, 11, 70, 3, 1, 1),       # K = 27 (pads to 32), Cout not a multiple of 64
    (1, 128, 120, 160, 256, 3, 1, 1),  # ≥1024 blocks of 128x128
]


@pytest.mark.parametrize("case", CASES)
def test_conv_bit_exact(ctx, case):
    """conv_max_split=1: a single k-ordered fmaf chain per output → bit-identical to the oracle."""
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    got = _run_conv(ctx, x, w, b, s, p, 0.1)
    ref = onet.conv2d(x, w, b, s, p, 0.1)
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    np.testing.assert_array_equal(got, ref)
    # default policy (auto split-K on under-filled grids): same sum re-associated across K slices
    got2 = _run_conv(ctx, x, w, b, s, p, 0.1)
    assert np.abs(got2 - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


DIRECT_CASES = [
    (2, 64, 60, 80, 128, 5, 2, 2),     # conv2 geometry (5x5 stride 2)
    (1, 128, 30, 40, 256, 5, 2, 2),    # conv3
    (2, 256, 15, 20, 256, 3, 1, 1),    # conv3_1 (3x3 stride 1), ragged last pixel tile
    (1, 512, 8, 10, 1024, 3, 2, 1),    # conv6: 20 output pixels in a 128-pixel tile
    (1, 6, 9, 11, 70, 3, 1, 1),        # K = 54 pads to 64; Cout not a multiple of 32
    (3, 10, 17, 13, 96, 7, 2, 3),      # 7x7 taps use validity bits ≥ 32
]


@pytest.mark.parametrize("case", DIRECT_CASES)
def test_conv_direct_kernel_bit_exact(ctx, case):
    """The LDS-free kernel (conv_direct=2: also without split-K) accumulates over (ci/2,ky,kx,ci%2): bit-identical to
    the oracle run in that order, and within fp32 re-association distance of the canonical order."""
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    lib.deepim_set_option(ctx.handle, b"conv_direct", 2)
    try:
        got = _run_conv(ctx, x, w, b, s, p, 0.1)
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
        lib.deepim_set_option(ctx.handle, b"conv_direct", 1)
    ref_pair = onet.conv2d(x, w, b, s, p, 0.1, pair_order=True)
    ref = onet.conv2d(x, w, b, s, p, 0.1)
    np.testing.assert_array_equal(got, ref_pair)
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    # default policy: LDS-free kernel + auto split-K
    got2 = _run_conv(ctx, x, w, b, s, p, 0.1)
    assert np.abs(got2 - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    # conv_direct = 0 falls back to the LDS kernel (canonical order when not split)
    lib.deepim_set_option(ctx.handle, b"conv_direct", 0)
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    try:
        got3 = _run_conv(ctx, x, w, b, s, p, 0.1)
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
        lib.deepim_set_option(ctx.handle, b"conv_direct", 1)
    np.testing.assert_array_equal(got3, ref)


def test_conv_direct_kernel_random_geometries(ctx):
    """Random small geometries (odd sizes, all strides/pads/kernel sizes the net uses, ragged tiles, Cout off the tile
    grid, split-K on and off): LDS-free kernel == pair-order oracle bit-for-bit, LDS kernel == canonical oracle."""
    rng = np.random.default_rng(2024)
    for trial in range(16):
        k = int(rng.choice([1, 3, 5, 7]))
        s_ = int(rng.choice([1, 2]))
        p_ = int(rng.integers(0, k // 2 + 1))
        cin = 2 * int(rng.integers(1, 20))
        cout = int(rng.choice([65, 96, 128, 130, 200, 256]))
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(k, 40)), int(rng.integers(k, 50))
        x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
        w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        case = (B, cin, H, W, cout, k, s_, p_)
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
        lib.deepim_set_option(ctx.handle, b"conv_direct", 2)
        try:
            got_d = _run_conv(ctx, x, w, b, s_, p_, 0.1)
            lib.deepim_set_option(ctx.handle, b"conv_direct", 0)
            got_l = _run_conv(ctx, x, w, b, s_, p_, 0.1)
        finally:
            lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
            lib.deepim_set_option(ctx.handle, b"conv_direct", 1)
        np.testing.assert_array_equal(got_d, onet.conv2d(x, w, b, s_, p_, 0.1, pair_order=True), err_msg=str(case))
        ref = onet.conv2d(x, w, b, s_, p_, 0.1)
        np.testing.assert_array_equal(got_l, ref, err_msg=str(case))
        got = _run_conv(ctx, x, w, b, s_, p_, 0.1)             # default policy
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), case


def _to_nc8(x):
    B, C, H, W = x.shape
    return np.ascontiguousarray(x.reshape(B, C // 8, 8, H, W).transpose(0, 1, 3, 4, 2))


def _from_nc8(y, shape):
    B, C, H, W = shape
    return np.ascontiguousarray(y.reshape(B, C // 8, H, W, 8).transpose(0, 1, 4, 2, 3).reshape(B, C, H, W))


NC8_CASES = [
    (2, 64, 60, 80, 128, 5, 2, 2),      # conv2 geometry
    (1, 128, 30, 40, 256, 3, 1, 1),     # 3x3 stride 1
    (3, 16, 17, 23, 72, 3, 1, 1),       # ragged pixel tile, Cout off the 128 grid (but % 8 == 0)
    (1, 512, 8, 10, 1024, 3, 2, 1),     # conv6: deep K, 20 output pixels
    (2, 8, 21, 19, 96, 7, 2, 3),        # 7x7 taps (validity bits >= 32), one channel block
    (1, 24, 9, 9, 136, 1, 1, 0),        # 1x1: K = 24 pads to 32 (two chunks of groups)
]


@pytest.mark.parametrize("case", NC8_CASES)
def test_conv_nc8_kernel(ctx, case):
    """Channel-blocked input ([n][C/8][h][w][8]) on the LDS-free kernel, NC8 and NCHW outputs: without split-K bit-identical
    to the oracle accumulating over (c/8, ky, kx, s, h), with split-K within fp32 re-association distance; the re-layout
    entry round-trips."""
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    ref2 = onet.conv2d(x, w, b, s, p, 0.1, pair_order=2)
    ref = onet.conv2d(x, w, b, s, p, 0.1)
    assert np.abs(ref2 - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    xin = ctx.empty(x.shape)
    lib.deepim_relayout_nc8(ctx.handle, xin, ctx.array(x), B, cin, H * W, 1)
    np.testing.assert_array_equal(xin.asnumpy().reshape(B, cin // 8, H, W, 8), _to_nc8(x))
    back = ctx.empty(x.shape)
    lib.deepim_relayout_nc8(ctx.handle, back, xin, B, cin, H * W, 0)
    np.testing.assert_array_equal(back.asnumpy(), x)
    pk, bias = _pack_conv(ctx, w), ctx.array(b)
    for max_split in (1, 0):
        lib.deepim_set_option(ctx.handle, b"conv_max_split", max_split)
        try:
            for out_nc8 in (1, 0):
                out = ctx.zeros((B, cout, Ho, Wo))
                lib.deepim_conv2d_forward_ex(ctx.handle, out, xin, pk, bias, B, cin, H, W, cout, k, k, s, p, cf(0.1), 0, 0, 1,
                                             out_nc8)
                got = _from_nc8(out.asnumpy(), (B, cout, Ho, Wo)) if out_nc8 else out.asnumpy()
                if max_split == 1:
                    np.testing.assert_array_equal(got, ref2, err_msg="out_nc8=%d" % out_nc8)
                else:
                    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
        finally:
            lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)


@pytest.mark.parametrize("case", [(2, 8, 96, 128, 64, 7, 2, 3), (1, 8, 37, 52, 64, 7, 2, 3), (3, 16, 20, 28, 64, 3, 1, 1),
                                  (1, 8, 480, 640, 64, 7, 2, 3)])
def test_conv_nc8_wide_kernel_for_64_output_channels(ctx, case):
    """conv1 on the channel-blocked net input: 64x256 tiles of the NC8 kernel (Cout = 64, NC8 output). Bit-identical to the
    oracle accumulating over (c/8, ky, kx, s, h); NCHW output is refused for this tile shape."""
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    ref2 = onet.conv2d(x, w, b, s, p, 0.1, pair_order=2)
    xin, pk, bias = ctx.array(_to_nc8(x)), _pack_conv(ctx, w), ctx.array(b)
    out = ctx.zeros((B, cout, Ho, Wo))
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    try:
        lib.deepim_conv2d_forward_ex(ctx.handle, out, xin, pk, bias, B, cin, H, W, cout, k, k, s, p, cf(0.1), 0, 0, 1, 1)
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    np.testing.assert_array_equal(_from_nc8(out.asnumpy(), (B, cout, Ho, Wo)), ref2)
    out2 = ctx.zeros((B, cout, Ho, Wo))                      # default plan (split-K allowed): same sums re-associated
    lib.deepim_conv2d_forward_ex(ctx.handle, out2, xin, pk, bias, B, cin, H, W, cout, k, k, s, p, cf(0.1), 0, 0, 1, 1)
    assert np.abs(_from_nc8(out2.asnumpy(), (B, cout, Ho, Wo)) - ref2).max() <= 1e-5 * max(1.0, np.abs(ref2).max())
    with pytest.raises(RuntimeError):
        lib.deepim_conv2d_forward_ex(ctx.handle, out, xin, pk, bias, B, cin, H, W, cout, k, k, s, p, cf(0.1), 0, 0, 1, 0)


def test_zoom_front_end_nc8_records_equal_the_nchw_tensor(ctx, small_batch):
    """deepim_zoom_concat_forward_nc8 writes (B,H,W,8) records whose elements are bit-identical to the (B,8,H,W) tensor of
    deepim_zoom_concat_forward, and the same zoom factor."""
    d = small_batch
    from mx_deepim_amd import synthetic
    B, H, W = 2, 480, 640
    args = [ctx.array(d["image_observed"]), ctx.array(d["image_rendered"][0]), ctx.array(d["mask_observed"]),
            ctx.array(d["mask_rendered"][0])]
    pose, K, means = ctx.array(d["src_pose"][0]), np.ascontiguousarray(d["K"]), np.ascontiguousarray(synthetic.PIXEL_MEANS[::-1])
    a, za = ctx.empty((B, 8, H, W)), ctx.empty((B, 4))
    lib.deepim_zoom_concat_forward(ctx.handle, args[0], args[1], args[2], args[3], None, None, pose, K, means, a, za, B, H, W)
    r, zr = ctx.empty((B, H, W, 8)), ctx.empty((B, 4))
    lib.deepim_zoom_concat_forward_nc8(ctx.handle, args[0], args[1], args[2], args[3], pose, K, means, r, zr, B, H, W)
    np.testing.assert_array_equal(zr.asnumpy(), za.asnumpy())
    np.testing.assert_array_equal(r.asnumpy().transpose(0, 3, 1, 2), a.asnumpy())


def test_conv_nc8_kernel_random_geometries(ctx):
    """Random geometries through the NC8 kernel (odd sizes, every kernel size / stride / pad the net uses, channel counts off
    the chunk grid so the K padding and the even-chunk rule of split-K are exercised, forced split factors)."""
    rng = np.random.default_rng(777)
    for trial in range(14):
        k = int(rng.choice([1, 3, 5, 7]))
        s_ = int(rng.choice([1, 2]))
        p_ = int(rng.integers(0, k // 2 + 1))
        cin = 8 * int(rng.integers(1, 9))
        cout = int(rng.choice([72, 96, 128, 136, 200, 256]))
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(k, 36)), int(rng.integers(k, 44))
        x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
        w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        Ho, Wo = (H + 2 * p_ - k) // s_ + 1, (W + 2 * p_ - k) // s_ + 1
        case = (B, cin, H, W, cout, k, s_, p_)
        ref2 = onet.conv2d(x, w, b, s_, p_, 0.1, pair_order=2)
        xin, pk, bias = ctx.array(_to_nc8(x)), _pack_conv(ctx, w), ctx.array(b)
        for plan in (1, 2, 3, 5):
            lib.deepim_set_option(ctx.handle, b"conv_force_plan", plan)
            try:
                out = ctx.zeros((B, cout, Ho, Wo))
                lib.deepim_conv2d_forward_ex(ctx.handle, out, xin, pk, bias, B, cin, H, W, cout, k, k, s_, p_, cf(0.1), 0, 0, 1,
                                             trial & 1)
            finally:
                lib.deepim_set_option(ctx.handle, b"conv_force_plan", 0)
            got = _from_nc8(out.asnumpy(), (B, cout, Ho, Wo)) if trial & 1 else out.asnumpy()
            if plan == 1:
                np.testing.assert_array_equal(got, ref2, err_msg=str(case))
            else:
                assert np.abs(got - ref2).max() <= 1e-5 * max(1.0, np.abs(ref2).max()), (case, plan)


def test_conv_nc8_argument_checks(ctx):
    x, pk = ctx.zeros((1, 12, 8, 8)), ctx.zeros((1 << 16,))
    out = ctx.zeros((1, 128, 8, 8))
    with pytest.raises(RuntimeError):   # Cin % 8 != 0 with NC8 input
        lib.deepim_conv2d_forward_ex(ctx.handle, out, x, pk, None, 1, 12, 8, 8, 128, 3, 3, 1, 1, cf(1.0), 0, 0, 1, 0)
    with pytest.raises(RuntimeError):   # NC8 output into a channel slice
        lib.deepim_conv2d_forward_ex(ctx.handle, out, x, pk, None, 1, 16, 8, 8, 64, 3, 3, 1, 1, cf(1.0), 128, 8, 0, 1)
    with pytest.raises(RuntimeError):   # NC8 input needs the 128-row tile
        lib.deepim_conv2d_forward_ex(ctx.handle, out, x, pk, None, 1, 16, 8, 8, 64, 3, 3, 1, 1, cf(1.0), 0, 0, 1, 0)
    with pytest.raises(RuntimeError):
        lib.deepim_relayout_nc8(ctx.handle, out, x, 1, 12, 64, 1)


def test_conv_tail_split_plan(ctx):
    """Opt-in tail split of the LDS-free kernel: 1030 tiles with a round size of 256 → 1024 full tiles + 6 tiles cut
    into K slices and summed by tail_reduce_kernel in slice order. Same sums re-associated: ≤1e-5 of the oracle."""
    rng = np.random.default_rng(5)
    B, cin, H, W, cout = 1, 32, 206, 320, 256          # npix = 65920 → 515 pixel tiles x 2 M tiles
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = onet.conv2d(x, w, b, 1, 1, 0.1)
    lib.deepim_set_option(ctx.handle, b"conv_tail_slots", 256)
    try:
        for plan in (-2, -3, -4, -18):
            lib.deepim_set_option(ctx.handle, b"conv_force_plan", plan)
            got = _run_conv(ctx, x, w, b, 1, 1, 0.1)
            assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), plan
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_force_plan", 0)
        lib.deepim_set_option(ctx.handle, b"conv_tail_slots", 1024)


@pytest.mark.parametrize("quad", [1, 0], ids=["four_pixels_per_lane", "one_pixel_per_lane"])
@pytest.mark.parametrize("case", [(2, 770, 30, 40, 2, 3, 1, 1), (1, 1026, 15, 20, 2, 3, 1, 1), (3, 37, 19, 23, 3, 5, 2, 2),
                                  (2, 5, 9, 70, 4, 7, 1, 3), (1, 3, 8, 8, 1, 1, 1, 0), (16, 64, 30, 40, 2, 3, 1, 1), (4, 1024, 8, 10, 2, 3, 1, 1),
                                  (2, 770, 30, 40, 1, 3, 1, 1), (3, 40, 6, 8, 3, 3, 1, 1), (2, 16, 5, 4, 4, 3, 1, 1), (5, 9, 1, 12, 2, 3, 1, 1)])
def test_conv_few_output_channels(ctx, case, quad):
    """Cout <= 4 (flow / mask heads) runs on the VALU streaming kernels by default: (ci,ky,kx)-ordered chains over channel shares,
    added in order (and, when the pixels alone would leave the chip empty, up to sixteen channel slices over grid.y with a fixed-order
    second pass) — within fp32 re-association distance of the canonical chain; writes into a channel slice too. The 3x3 stride-1 pad-1
    heads with W % 4 == 0 take the four-pixels-per-lane form (rows of one quad, one-row images, Cout = 1 … 4 among the cases)."""
    lib.deepim_set_option(ctx.handle, b"conv_fewout_quad", quad)
    try:
        _few_output_channels(ctx, case)
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_fewout_quad", 1)


def _few_output_channels(ctx, case):
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = onet.conv2d(x, w, b, s, p, 0.1)
    got = _run_conv(ctx, x, w, b, s, p, 0.1)
    assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    Ho, Wo = ref.shape[2:]
    out = ctx.zeros((B, cout + 5, Ho, Wo))
    lib.deepim_conv2d_forward(ctx.handle, out, ctx.array(x), _pack_conv(ctx, w), ctx.array(b), B, cin, H, W, cout, k, k, s, p,
                              cf(0.1), cout + 5, 3)
    o = out.asnumpy()
    np.testing.assert_array_equal(o[:, 3:3 + cout], got)
    assert not o[:, :3].any() and not o[:, 3 + cout:].any()


def test_conv_matches_torch_cpu(ctx):
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(11)
    x = rng.standard_normal((2, 8, 64, 80)).astype(np.float32)
    w = (rng.standard_normal((64, 8, 7, 7)) / 20).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    got = _run_conv(ctx, x, w, b, 2, 3, 0.1)
    t = torch.nn.functional.leaky_relu(
        torch.nn.functional.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=2, padding=3), 0.1)
    np.testing.assert_allclose(got, t.numpy(), rtol=1e-4, atol=1e-5)


def test_conv_channel_slice_output(ctx):
    rng = np.random.default_rng(12)
    x = rng.standard_normal((2, 8, 20, 24)).astype(np.float32)
    w = rng.standard_normal((64, 8, 3, 3)).astype(np.float32)
    out = ctx.zeros((2, 100, 20, 24))
    lib.deepim_conv2d_forward(ctx.handle, out, ctx.array(x), _pack_conv(ctx, w), None, 2, 8, 20, 24, 64, 3, 3, 1, 1,
                              cf(1.0), 100, 30)
    got = out.asnumpy()
    np.testing.assert_array_equal(got[:, 30:94], onet.conv2d(x, w, None, 1, 1, 1.0))
    assert not got[:, :30].any() and not got[:, 94:].any()


@pytest.mark.parametrize("case", [(2, 1024, 8, 10, 512, 15, 20), (2, 1026, 15, 20, 256, 30, 40), (2, 2, 8, 10, 2, 15, 20),
                                  (1, 5, 7, 9, 3, 15, 19)])
def test_deconv_crop_bit_exact(ctx, case):
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    B, cin, H, W, cout, ho, wo = case
    rng = np.random.default_rng(7 + cin)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, 4, 4)) / np.sqrt(cin * 4)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    pk = DeviceArray(ctx, (lib.load().deepim_deconv_packed_size(cin, cout) // 4,))
    lib.deepim_deconv_pack_weights(ctx.handle, pk, ctx.array(w), cin, cout)
    out = ctx.empty((B, cout, ho, wo))
    lib.deepim_deconv4x4s2_crop_forward(ctx.handle, out, ctx.array(x), pk, ctx.array(b), B, cin, H, W, cout, ho, wo, 1, 1,
                                        cf(0.1), 0, 0)
    ref = onet.deconv4x4s2_crop(x, w, b, ho, wo, (1, 1), 0.1)
    got = out.asnumpy()
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    np.testing.assert_array_equal(got, ref)
    lib.deepim_deconv4x4s2_crop_forward(ctx.handle, out, ctx.array(x), pk, ctx.array(b), B, cin, H, W, cout, ho, wo, 1, 1,
                                        cf(0.1), 0, 0)
    assert np.abs(out.asnumpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    # default plan again, into a channel slice of a wider tensor (the decoder's Concat): even Cin with Cout >= 64 takes the grouped
    # register-fed kernel (round 4: four parity-class convolutions in one launch, weights packed once), its split-K second pass
    # included at these sizes; everything outside the slice stays untouched
    cat = ctx.array(np.full((B, cout + 5, ho, wo), 7.0, np.float32))
    lib.deepim_deconv4x4s2_crop_forward(ctx.handle, cat, ctx.array(x), pk, ctx.array(b), B, cin, H, W, cout, ho, wo, 1, 1,
                                        cf(0.1), cout + 5, 3)
    got = cat.asnumpy()
    assert np.abs(got[:, 3:3 + cout] - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
    assert (got[:, :3] == 7.0).all() and (got[:, 3 + cout:] == 7.0).all()


def test_upsample16_crop(ctx):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 2, 30, 40)).astype(np.float32)
    w = onet.bilinear_upsample_weights(2)
    out = ctx.empty((2, 2, 480, 640))
    lib.deepim_upsample16_crop_forward(ctx.handle, out, ctx.array(x), ctx.array(w), 2, 2, 30, 40, 480, 640, 8, 8, cf(20.0))
    np.testing.assert_array_equal(out.asnumpy(), onet.upsample16_crop(x, w, 480, 640, (8, 8), 20.0))      # four outputs per thread
    out2 = ctx.empty((2, 2, 470, 638))      # a width / crop the 16-byte form does not take: the one-output kernel, the same bits
    lib.deepim_upsample16_crop_forward(ctx.handle, out2, ctx.array(x), ctx.array(w), 2, 2, 30, 40, 470, 638, 5, 9, cf(1.0))
    np.testing.assert_array_equal(out2.asnumpy(), onet.upsample16_crop(x, w, 470, 638, (5, 9), 1.0))


@pytest.mark.parametrize("shape", [(16, 81920, 256), (3, 256, 256), (2, 256, 4), (17, 1024, 40)])
def test_fc(ctx, shape):
    B, I, O = shape
    rng = np.random.default_rng(B)
    x = rng.standard_normal((B, I)).astype(np.float32)
    w = (rng.standard_normal((O, I)) / np.sqrt(I)).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32)
    out = ctx.empty((B, O))
    lib.deepim_fc_forward(ctx.handle, out, ctx.array(x), ctx.array(w), ctx.array(b), B, I, O, cf(0.1))
    ref = onet.fc(x, w, b, 0.1)
    np.testing.assert_allclose(out.asnumpy(), ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("shape", [(32, 81920, 256), (16, 81920, 256), (1, 81920, 256), (40, 81920, 256), (5, 256, 256),
                                   (3, 1000 * 8, 256)])
def test_fc_packed_mfma(ctx, shape):
    """fc6 on the matrix cores (one weight pass per 32 batch rows, fixed-order split-K reduction) vs the oracle, and
    bit-reproducible from call to call."""
    B, I, O = shape
    rng = np.random.default_rng(B + I)
    x = rng.standard_normal((B, I)).astype(np.float32)
    w = (rng.standard_normal((O, I)) / np.sqrt(I)).astype(np.float32)
    b = rng.standard_normal(O).astype(np.float32)
    pk = DeviceArray(ctx, (lib.load().deepim_fc_packed_size(O, I) // 4,))
    lib.deepim_fc_pack_weights(ctx.handle, pk, ctx.array(w), O, I)
    out = ctx.empty((B, O))
    dx, db = ctx.array(x), ctx.array(b)
    lib.deepim_fc_forward_packed(ctx.handle, out, dx, pk, db, B, I, O, cf(0.1))
    got = out.asnumpy()
    np.testing.assert_allclose(got, onet.fc(x, w, b, 0.1), rtol=1e-5, atol=1e-5)
    lib.deepim_fc_forward_packed(ctx.handle, out, dx, pk, db, B, I, O, cf(0.1))
    np.testing.assert_array_equal(out.asnumpy(), got)
    with pytest.raises(RuntimeError):
        lib.deepim_fc_forward_packed(ctx.handle, out, dx, pk, db, B, I, 128, cf(0.1))


def test_copy_channels(ctx):
    rng = np.random.default_rng(4)
    src = rng.standard_normal((3, 5, 6, 7)).astype(np.float32)
    dst = ctx.zeros((3, 9, 6, 7))
    lib.deepim_copy_channels(ctx.handle, dst, 9, 2, ctx.array(src), 5, 3, 42)
    got = dst.asnumpy()
    np.testing.assert_array_equal(got[:, 2:7], src)
    assert not got[:, :2].any() and not got[:, 7:].any()


def test_conv_input_over_2gib_runs_as_sub_batches(ctx):
    """A conv2-shaped launch whose input tensor exceeds 2 GiB (B = 112 at 64x240x320; the raw-buffer descriptor of one
    launch addresses < 2 GiB) is run as consecutive sub-batches: every sample equals the same sample run alone."""
    rng = np.random.default_rng(21)
    B, cin, H, W, cout, k, s, pd = 112, 64, 240, 320, 128, 5, 2, 2
    assert B * cin * H * W * 4 > 2 ** 31
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32)
    pk, db = _pack_conv(ctx, w), ctx.array(bias)
    base = rng.standard_normal((4, cin, H, W)).astype(np.float32)
    x = ctx.empty((B, cin, H, W))
    per = cin * H * W * 4
    for b in range(B):   # samples cycle through 4 patterns, scaled so every sample differs
        lib.deepim_h2d(ctx.handle, ctypes.c_void_p(x.ptr + b * per), np.ascontiguousarray(base[b % 4] * np.float32(1 + b // 4)), per)
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)   # one canonical chain per output: independent of B
    try:
        ho, wo = 120, 160
        out = ctx.empty((B, cout, ho, wo))
        lib.deepim_conv2d_forward(ctx.handle, out, x, pk, db, B, cin, H, W, cout, k, k, s, pd, cf(0.1), 0, 0)
        one = ctx.empty((1, cout, ho, wo))
        full = out.asnumpy()
        for b in (0, 53, 54, 108, 109, 111):   # both sides of the sub-batch boundaries (108 samples fit in 2 GiB)
            xb = ctx.array(np.ascontiguousarray(base[b % 4] * np.float32(1 + b // 4))[None])
            lib.deepim_conv2d_forward(ctx.handle, one, xb, pk, db, 1, cin, H, W, cout, k, k, s, pd, cf(0.1), 0, 0)
            np.testing.assert_array_equal(full[b], one.asnumpy()[0])
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)


def test_default_split_k_plan_is_deterministic(ctx):
    """The default plan comes from a cost model, not from timings: two fresh contexts' worth of calls (here: the same
    geometry launched repeatedly, and after an unrelated geometry) give bit-identical split-K results."""
    rng = np.random.default_rng(22)
    x = rng.standard_normal((2, 512, 15, 20)).astype(np.float32)
    w = (rng.standard_normal((512, 512, 3, 3)) / 68).astype(np.float32)
    a = _run_conv(ctx, x, w, None, 1, 1, 0.1)
    _run_conv(ctx, rng.standard_normal((1, 8, 40, 40)).astype(np.float32), (rng.standard_normal((64, 8, 3, 3)) / 8).astype(np.float32), None, 1, 1, 0.1)
    b = _run_conv(ctx, x, w, None, 1, 1, 0.1)
    np.testing.assert_array_equal(a, b)
    ref = onet.conv2d(x, w, None, 1, 1, 0.1)
    assert np.abs(a - ref).max() <= 1e-5 * np.abs(ref).max()
