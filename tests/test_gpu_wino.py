"""fp32 Winograd F(2x2,3x3) layers (csrc/wino.hip) against the C oracle's direct convolution: a different summation of the same
fp32 products, so the bar is a tolerance — 1e-5 of the layer's output range — not bit equality."""
import ctypes

import numpy as np
import pytest

from oracle import net as onet
from mx_deepim_amd.runtime import DeviceArray, lib

pytestmark = pytest.mark.gpu
cf = ctypes.c_float
TOL = 1e-5


@pytest.fixture(params=["shared", "shared_wide", "shared_half", "one_wave", "two_wave"],
                ids=["shared_transform_64x64", "shared_transform_128x32", "shared_transform_64x32_two_blocks_per_cu", "one_wave_per_simd",
                     "two_waves_per_simd"], autouse=True)
def wino_kernel(ctx, request):
    """Every test on all kernels: the 8-wave shared-transform blocks (64 channels x 64 tiles wherever Cout % 64 == 0, 128 x 32 wherever
    Cout % 128 == 0, 64 x 32 on four waves with two blocks per CU; the one-wave kernel otherwise), the round-4 one-wave kernel alone, and the optional two-wave form."""
    lib.deepim_set_option(ctx.handle, b"wino_shared", 1 if request.param.startswith("shared") else 0)
    lib.deepim_set_option(ctx.handle, b"wino_wide", {"shared_wide": 3, "shared_half": 2}.get(request.param, 0))
    lib.deepim_set_option(ctx.handle, b"wino_two_wave", 1 if request.param == "two_wave" else 0)
    yield "shared" if request.param.startswith("shared") else request.param
    lib.deepim_set_option(ctx.handle, b"wino_two_wave", 0)
    lib.deepim_set_option(ctx.handle, b"wino_wide", 1)
    lib.deepim_set_option(ctx.handle, b"wino_shared", 1)
    lib.deepim_set_option(ctx.handle, b"wino_streamk", 1)
    lib.deepim_set_option(ctx.handle, b"wino_split", 0)
    lib.deepim_set_option(ctx.handle, b"wino_fin", 0)


def _to_nc8(x):
    B, C, H, W = x.shape
    return np.ascontiguousarray(x.reshape(B, C // 8, 8, H, W).transpose(0, 1, 3, 4, 2))


def _from_nc8(y, shape):
    B, C, H, W = shape
    return np.ascontiguousarray(y.reshape(B, C // 8, H, W, 8).transpose(0, 1, 4, 2, 3).reshape(B, C, H, W))


def _pack(ctx, w):
    cout, cin = w.shape[:2]
    pk = DeviceArray(ctx, (lib.load().deepim_conv_wino_packed_size(cout, cin) // 4,))
    lib.deepim_conv_wino_pack_weights(ctx.handle, pk, ctx.array(w), cout, cin)
    return pk


# (B, Cin, H, W, Cout): the four encoder geometries at reduced size, odd H / W (half-covered last tiles), one-tile images,
# a tile count off the 128 grid, several channel blocks per XCD slice and a channel-block count that is not a multiple of 8
CASES = [
    (2, 256, 12, 16, 256),     # conv3_1 channels
    (1, 512, 30, 40, 512),     # conv4_1 at full spatial size, one sample
    (3, 64, 15, 20, 64),       # conv5_1 geometry (odd H), few channels
    (2, 1024, 8, 10, 1024),    # conv6_1
    (5, 8, 7, 9, 32),          # odd H and W, one body pair, one channel block
    (1, 16, 2, 2, 96),         # a single tile, 3 channel blocks
    (2, 24, 1, 5, 160),        # one row: every tile is half outside
    (33, 8, 6, 6, 32),         # 297 tiles: ragged last block of 128
]


@pytest.mark.parametrize("case", CASES)
def test_wino_layer_within_1e5_of_the_direct_convolution(ctx, case):
    B, cin, H, W, cout = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    x *= (rng.uniform(size=x.shape) > 0.3)          # post-LeakyReLU-like sparsity does not matter, zeros must survive exactly
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = onet.conv2d(x, w, b, 1, 1, 0.1)
    scale = max(1.0, float(np.abs(ref).max()))
    xin, pk, bias = ctx.array(_to_nc8(x)), _pack(ctx, w), ctx.array(b)
    out = ctx.zeros((B, cout, H, W))
    lib.deepim_conv2d_wino_forward(ctx.handle, out, xin, pk, bias, B, cin, H, W, cout, cf(0.1), 1, 0, 0)
    got = _from_nc8(out.asnumpy(), (B, cout, H, W))
    assert np.abs(got - ref).max() <= TOL * scale, np.abs(got - ref).max() / scale
    # NCHW output into a channel slice of a wider tensor, no bias, no activation
    wide = ctx.array(np.full((B, cout + 5, H, W), 7.0, np.float32))
    lib.deepim_conv2d_wino_forward(ctx.handle, wide, xin, pk, None, B, cin, H, W, cout, cf(1.0), 0, cout + 5, 3)
    ref0 = onet.conv2d(x, w, np.zeros(cout, np.float32), 1, 1, 1.0)
    gw = wide.asnumpy()
    assert np.abs(gw[:, 3:3 + cout] - ref0).max() <= TOL * max(1.0, float(np.abs(ref0).max()))
    assert (gw[:, :3] == 7.0).all() and (gw[:, 3 + cout:] == 7.0).all()


@pytest.mark.parametrize("case", [(2, 256, 12, 16, 256), (3, 64, 15, 20, 64), (2, 24, 1, 5, 128), (33, 8, 6, 6, 64), (1, 1024, 8, 10, 1024)])
def test_shared_transform_kernel_is_bit_identical_to_the_one_wave_kernel(ctx, case, wino_kernel):
    """conv_wino8_kernel computes the same V, the same per-position fp32 MFMA chains and the same output-transform expression tree
    as conv_wino_kernel — only who computes what differs — so NC8, space-to-depth and NCHW-slice outputs are the same bits."""
    if wino_kernel != "shared":
        pytest.skip("compares the two kernels itself")
    lib.deepim_set_option(ctx.handle, b"wino_split", 1)      # one block walks all input channels: the same summation order
    B, cin, H, W, cout = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    xin, pk, bias = ctx.array(_to_nc8(x)), _pack(ctx, w), ctx.array(rng.standard_normal(cout).astype(np.float32))
    modes = [(1, 0, 0)] + ([(3, 0, 0)] if H % 2 == 0 and W % 2 == 0 else []) + [(0, cout + 8, 8)]
    for out_nc8, ctotal, coff in modes:
        outs = []
        for shared in (1, 0):
            lib.deepim_set_option(ctx.handle, b"wino_shared", shared)
            o = ctx.array(np.full((B, max(ctotal, cout), H, W), 3.0, np.float32))
            lib.deepim_conv2d_wino_forward(ctx.handle, o, xin, pk, bias, B, cin, H, W, cout, cf(0.1), out_nc8, ctotal, coff)
            outs.append(o.asnumpy())
        lib.deepim_set_option(ctx.handle, b"wino_shared", 1)
        np.testing.assert_array_equal(outs[0], outs[1])
    lib.deepim_set_option(ctx.handle, b"wino_split", 0)


@pytest.mark.parametrize("case", [(2, 256, 12, 16, 256), (1, 1024, 8, 10, 1024), (3, 64, 15, 20, 128), (4, 512, 15, 20, 512)])
@pytest.mark.parametrize("slices", [0, 2, 3])
def test_shared_transform_kernel_split_over_the_input_channels(ctx, case, slices, wino_kernel):
    """Under-filled grids split the input channels over several blocks (raw sums per slice, second pass adds bias + LeakyReLU):
    the planner's own choice (0) and forced slice counts, NC8 / space-to-depth / NCHW-slice outputs, against the unsplit kernel."""
    if wino_kernel != "shared":
        pytest.skip("the shared-transform kernel's own path")
    B, cin, H, W, cout = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    xin, pk, bias = ctx.array(_to_nc8(x)), _pack(ctx, w), ctx.array(rng.standard_normal(cout).astype(np.float32))
    modes = [(1, 0, 0)] + ([(3, 0, 0)] if H % 2 == 0 and W % 2 == 0 else []) + [(0, cout + 8, 8)]
    for out_nc8, ctotal, coff in modes:
        outs = []
        for split in (1, slices):
            lib.deepim_set_option(ctx.handle, b"wino_split", split)
            o = ctx.array(np.full((B, max(ctotal, cout), H, W), 3.0, np.float32))
            lib.deepim_conv2d_wino_forward(ctx.handle, o, xin, pk, bias, B, cin, H, W, cout, cf(0.1), out_nc8, ctotal, coff)
            outs.append(o.asnumpy())
        # the slices summed by the slice that arrives last (wino_fin = 1) instead of by the second pass (the default): the same adds in the
        # same order — bit for bit; twice, so that the arrival counters are seen to return to zero
        lib.deepim_set_option(ctx.handle, b"wino_split", slices)
        lib.deepim_set_option(ctx.handle, b"wino_fin", 1)
        o = ctx.array(np.full((B, max(ctotal, cout), H, W), 3.0, np.float32))
        lib.deepim_conv2d_wino_forward(ctx.handle, o, xin, pk, bias, B, cin, H, W, cout, cf(0.1), out_nc8, ctotal, coff)
        np.testing.assert_array_equal(o.asnumpy(), outs[1])
        o2 = ctx.array(np.full((B, max(ctotal, cout), H, W), 3.0, np.float32))
        lib.deepim_conv2d_wino_forward(ctx.handle, o2, xin, pk, bias, B, cin, H, W, cout, cf(0.1), out_nc8, ctotal, coff)
        np.testing.assert_array_equal(o2.asnumpy(), outs[1])
        lib.deepim_set_option(ctx.handle, b"wino_fin", 0)
        lib.deepim_set_option(ctx.handle, b"wino_split", 0)
        scale = max(1.0, float(np.abs(outs[0]).max()))
        assert np.abs(outs[0] - outs[1]).max() <= 2e-6 * scale
        if out_nc8 == 0:
            assert (outs[1][:, :coff] == 3.0).all()


def test_wino_weight_transform_is_G_g_Gt_with_the_last_column_negated(ctx):
    rng = np.random.default_rng(5)
    cout, cin = 32, 16
    w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    pk = _pack(ctx, w).asnumpy().reshape(cout // 32, cin // 8, 16, 2, 32, 4)     # [mb][c8][pos][h][row][s]
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    U = np.einsum("xa,ocab,nb->ocxn", G, w.astype(np.float64), G).astype(np.float32)   # (cout, cin, 4, 4)
    U[..., 3] = -U[..., 3]      # positions nu = 3 are stored negated (the kernels multiply them with t3 - t1 instead of t1 - t3)
    for c8 in range(cin // 8):
        for h in range(2):
            for s in range(4):
                np.testing.assert_array_equal(pk[0, c8, :, h, :, s], U[:, c8 * 8 + 4 * h + s].reshape(cout, 16).T)


def test_wino_argument_checks(ctx):
    assert lib.load().deepim_conv_wino_packed_size(48, 8) == 0 and lib.load().deepim_conv_wino_packed_size(32, 12) == 0
    d = ctx.zeros((64,))
    with pytest.raises(RuntimeError):
        lib.deepim_conv2d_wino_forward(ctx.handle, d, d, d, None, 1, 12, 2, 2, 32, cf(0.1), 1, 0, 0)
    with pytest.raises(RuntimeError):
        lib.deepim_conv2d_wino_forward(ctx.handle, d, d, d, None, 1, 8, 2, 2, 40, cf(0.1), 1, 0, 0)
    lib.deepim_conv2d_wino_forward(ctx.handle, d, d, d, None, 0, 8, 2, 2, 32, cf(0.1), 1, 0, 0)      # empty batch: no launch

def _plan(ctx, B, cin, H, W, cout, out_nc8, s2d):
    plan = (ctypes.c_int * 9)()
    assert lib.load().deepim_conv_wino_plan(ctx.handle, B, cin, H, W, cout, out_nc8, s2d, plan) == 0
    return list(plan)


# (B, Cin, H, W, Cout, forced): grids beyond one round of resident blocks on every block shape; forced = 2: stream-K wherever it applies
# (few K steps: the cost model itself would not cut these), 1: the cost model's own choice (conv3_1 at the per-GPU share of an 8-GPU node)
STREAMK_CASES = [
    (4, 32, 96, 128, 128, 2),      # two granules per tile block, one channel block (wide) / two
    (9, 64, 40, 56, 256, 2),       # ragged last tile block, padded grid (invalid tile blocks inside the runs)
    (4, 256, 60, 80, 256, 1),      # conv3_1, B = 4
]


@pytest.mark.parametrize("case", STREAMK_CASES)
def test_shared_transform_kernel_stream_k(ctx, case, wino_kernel):
    """Grids that end in a partly filled round: the persistent blocks share that round granule by granule, a cut tile block's raw copies
    are summed by whichever piece arrives last. Against the whole-tile-block walk (same kernel, wino_streamk = 0): rounding of the K cuts;
    twice the same bits (the order of the adds is fixed, not the order of arrival)."""
    if wino_kernel != "shared":
        pytest.skip("the shared-transform kernel's own path")
    B, cin, H, W, cout, forced = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    xin, pk, bias = ctx.array(_to_nc8(x)), _pack(ctx, w), ctx.array(b)
    for out_nc8 in (1, 3):
        outs = []
        for sk in (0, forced):
            lib.deepim_set_option(ctx.handle, b"wino_streamk", sk)
            plan = _plan(ctx, B, cin, H, W, cout, out_nc8, 0)
            if sk and plan[4] == 0 and case != STREAMK_CASES[0]:
                pytest.skip("the last round of this grid on this block shape has too few granules per block: no stream-K")
            assert (plan[4] > 0) == (sk > 0), plan
            if sk:
                assert plan[2] == 1 and plan[5] >= 1 and plan[6] >= 1 and plan[1] in (256, 512), plan
            o = ctx.array(np.full((B, cout, H, W), 3.0, np.float32))
            lib.deepim_conv2d_wino_forward(ctx.handle, o, xin, pk, bias, B, cin, H, W, cout, cf(0.1), out_nc8, 0, 0)
            outs.append(o.asnumpy())
        scale = max(1.0, float(np.abs(outs[0]).max()))
        assert np.abs(outs[0] - outs[1]).max() <= 2e-6 * scale
        assert not (outs[1] == 3.0).any()
        o = ctx.array(np.full((B, cout, H, W), 5.0, np.float32))
        lib.deepim_conv2d_wino_forward(ctx.handle, o, xin, pk, bias, B, cin, H, W, cout, cf(0.1), out_nc8, 0, 0)
        np.testing.assert_array_equal(o.asnumpy(), outs[1])
    # NCHW output: stream-K does not apply (the plan says so), the layer still runs
    lib.deepim_set_option(ctx.handle, b"wino_streamk", 2)
    assert _plan(ctx, B, cin, H, W, cout, 0, 0)[4] == 0
    if case == STREAMK_CASES[0]:
        ref = onet.conv2d(x, w, b, 1, 1, 0.1)
        o = ctx.zeros((B, cout, H, W))
        lib.deepim_conv2d_wino_forward(ctx.handle, o, xin, pk, bias, B, cin, H, W, cout, cf(0.1), 1, 0, 0)
        got = _from_nc8(o.asnumpy(), (B, cout, H, W))
        assert np.abs(got - ref).max() <= TOL * max(1.0, float(np.abs(ref).max()))


def test_stride2_layer_stream_k(ctx, wino_kernel):
    """The 5x5 stride-2 form cuts between its eight-step loop bodies (granule = 8 steps = two 8-channel blocks of each input phase)."""
    if wino_kernel != "shared":
        pytest.skip("the shared-transform kernel's own path")
    B, cin, H, W, cout = 6, 32, 128, 128, 256
    rng = np.random.default_rng(5)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 5, 5)) / np.sqrt(cin * 25)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    Ho, Wo = H // 2, W // 2
    xs = ctx.empty((B, 4 * cin, Ho, Wo))
    lib.deepim_relayout_nc8_s2d(ctx.handle, xs, ctx.array(x), B, cin, H, W, 1)
    pk = DeviceArray(ctx, (lib.load().deepim_conv_wino_packed_size(cout, 4 * cin) // 4,))
    lib.deepim_conv_wino_pack_weights_s2d(ctx.handle, pk, ctx.array(w), cout, cin)
    outs = []
    for sk in (0, 2):
        lib.deepim_set_option(ctx.handle, b"wino_streamk", sk)
        plan = _plan(ctx, B, 4 * cin, Ho, Wo, cout, 3, 1)
        assert (plan[4] == 2) == (sk > 0), plan            # 16 steps = 2 granules of 8
        o = ctx.zeros((B, cout, Ho, Wo))
        lib.deepim_conv2d_wino_forward_s2d(ctx.handle, o, xs, pk, ctx.array(b), B, cin, H, W, cout, cf(0.1), 3, 0, 0)
        nchw = ctx.empty((B, cout, Ho, Wo))
        lib.deepim_relayout_nc8_s2d(ctx.handle, nchw, o, B, cout, Ho, Wo, 0)
        outs.append(nchw.asnumpy())
    ref = onet.conv2d(x, w, b, 2, 2, 0.1)
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(outs[1] - ref).max() <= TOL * scale
    assert np.abs(outs[1] - outs[0]).max() <= 2e-6 * scale


S2D_CASES = [
    (2, 64, 24, 32, 128),      # conv2 channels at reduced size
    (1, 128, 12, 16, 256),     # conv3 channels
    (3, 8, 6, 10, 32),         # one channel block per phase
    (1, 16, 2, 2, 64),         # a single output pixel row / column pair: every tap but the centre ones in the padding
    (2, 32, 10, 14, 96),       # four blocks of 8 channels per phase: the phase loops run their steady-state bodies
    (2, 32, 12, 16, 64),       # the same on the shared-transform kernel (Cout % 64 == 0): two eight-step loop bodies
    (3, 16, 6, 10, 128),       # one loop body, ragged tile block
]


@pytest.mark.parametrize("case", S2D_CASES)
def test_wino_stride2_5x5_layer_over_space_to_depth_input(ctx, case, wino_kernel):
    """conv2 / conv3 (5x5, stride 2, pad 2) as the 3x3 stride-1 problem over the four input phases: relayout round trip, the layer
    within 1e-5 of the direct convolution, and its own output in space-to-depth order (what the next such layer reads)."""
    B, cin, H, W, cout = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 5, 5)) / np.sqrt(cin * 25)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = onet.conv2d(x, w, b, 2, 2, 0.1)
    Ho, Wo = H // 2, W // 2
    assert ref.shape == (B, cout, Ho, Wo)
    xs = ctx.empty((B, 4 * cin, Ho, Wo))
    lib.deepim_relayout_nc8_s2d(ctx.handle, xs, ctx.array(x), B, cin, H, W, 1)
    want = np.concatenate([x[:, :, py::2, px::2] for py in (0, 1) for px in (0, 1)], axis=1)      # channel (py*2+px)*cin + c
    np.testing.assert_array_equal(_from_nc8(xs.asnumpy(), (B, 4 * cin, Ho, Wo)), want)
    back = ctx.empty(x.shape)
    lib.deepim_relayout_nc8_s2d(ctx.handle, back, xs, B, cin, H, W, 0)
    np.testing.assert_array_equal(back.asnumpy(), x)
    pk = DeviceArray(ctx, (lib.load().deepim_conv_wino_packed_size(cout, 4 * cin) // 4,))
    lib.deepim_conv_wino_pack_weights_s2d(ctx.handle, pk, ctx.array(w), cout, cin)
    out = ctx.zeros((B, cout, Ho, Wo))
    lib.deepim_conv2d_wino_forward(ctx.handle, out, xs, pk, ctx.array(b), B, 4 * cin, Ho, Wo, cout, cf(0.1), 1, 0, 0)
    got = _from_nc8(out.asnumpy(), (B, cout, Ho, Wo))
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(got - ref).max() <= TOL * scale, np.abs(got - ref).max() / scale
    # the layer's own entry: same kernel with the identically-zero positions of the odd phases skipped (Cin % 16 == 0; otherwise the
    # full loop) — the skipped terms are exact zeros, so the results are the same bits
    out_s = ctx.zeros((B, cout, Ho, Wo))
    lib.deepim_conv2d_wino_forward_s2d(ctx.handle, out_s, xs, pk, ctx.array(b), B, cin, H, W, cout, cf(0.1), 1, 0, 0)
    got_s = _from_nc8(out_s.asnumpy(), (B, cout, Ho, Wo))
    if wino_kernel == "shared" and cout % 64 == 0 and (4 * cin) % 64 == 0:
        # the shared-transform kernel walks the four input phases interleaved (a compile-time phase per step instead of branches):
        # the same products summed in another channel order
        assert np.abs(got_s - ref).max() <= TOL * scale
        assert np.abs(got_s - got).max() <= 2e-6 * scale
    else:
        np.testing.assert_array_equal(got_s, got)
    lib.deepim_set_option(ctx.handle, b"wino_s2d_skip", 0)
    try:
        out_f = ctx.zeros((B, cout, Ho, Wo))
        lib.deepim_conv2d_wino_forward_s2d(ctx.handle, out_f, xs, pk, ctx.array(b), B, cin, H, W, cout, cf(0.1), 0, 0, 0)
    finally:
        lib.deepim_set_option(ctx.handle, b"wino_s2d_skip", 1)
    np.testing.assert_array_equal(out_f.asnumpy(), got)
    if Ho % 2 == 0 and Wo % 2 == 0:
        out2 = ctx.zeros((B, cout, Ho, Wo))
        lib.deepim_conv2d_wino_forward(ctx.handle, out2, xs, pk, ctx.array(b), B, 4 * cin, Ho, Wo, cout, cf(0.1), 3, 0, 0)
        nchw = ctx.empty((B, cout, Ho, Wo))
        lib.deepim_relayout_nc8_s2d(ctx.handle, nchw, out2, B, cout, Ho, Wo, 0)
        np.testing.assert_array_equal(nchw.asnumpy(), got)


@pytest.mark.parametrize("max_split", [1, 0])
def test_direct_kernels_write_space_to_depth_output(ctx, max_split):
    """out_nc8 = 3 of deepim_conv2d_forward_ex: the same values as the NC8 output, at their space-to-depth addresses — from the
    NCHW-input kernel (conv1's) and from the NC8-input kernel, with and without a split-K second pass."""
    rng = np.random.default_rng(3)
    for (B, cin, H, W, cout, k, s, p, in8) in [(2, 8, 24, 40, 64, 7, 2, 3, 0), (1, 64, 12, 20, 128, 5, 2, 2, 1), (1, 512, 8, 12, 256, 3, 1, 1, 1)]:
        x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
        w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
        pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cout, cin, k, k) // 4,))
        lib.deepim_conv_pack_weights(ctx.handle, pk, ctx.array(w), cout, cin, k, k)
        xin = ctx.array(_to_nc8(x) if in8 else x)
        lib.deepim_set_option(ctx.handle, b"conv_max_split", max_split)
        try:
            o1, o3 = ctx.zeros((B, cout, Ho, Wo)), ctx.zeros((B, cout, Ho, Wo))
            lib.deepim_conv2d_forward_ex(ctx.handle, o1, xin, pk, ctx.array(b), B, cin, H, W, cout, k, k, s, p, cf(0.1), 0, 0, in8, 1)
            lib.deepim_conv2d_forward_ex(ctx.handle, o3, xin, pk, ctx.array(b), B, cin, H, W, cout, k, k, s, p, cf(0.1), 0, 0, in8, 3)
        finally:
            lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
        nchw = ctx.empty((B, cout, Ho, Wo))
        lib.deepim_relayout_nc8_s2d(ctx.handle, nchw, o3, B, cout, Ho, Wo, 0)
        np.testing.assert_array_equal(nchw.asnumpy(), _from_nc8(o1.asnumpy(), (B, cout, Ho, Wo)))
