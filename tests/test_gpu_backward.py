"""§8f-4 — backward of the conv stack / FC head, SGD, and one training-style iteration of the pose branch on the GPU against
the oracle backward (oracle/net.c, float64 accumulation; itself checked against torch autograd in
tests/test_oracle_thirdparty.py). Tolerances: gradients are long fp32 sums in a different order from the float64 oracle, so
1e-4 of the tensor's max magnitude (observed ~1e-6)."""
import ctypes

import numpy as np
import pytest

from oracle import net as onet
from oracle import pipeline as opipe
from mx_deepim_amd import synthetic
from mx_deepim_amd.config import default_config
from mx_deepim_amd.runtime import DeviceArray, lib
from mx_deepim_amd.symbols import deepIM_flownet

pytestmark = pytest.mark.gpu
cf = ctypes.c_float
MEANS_REV = np.ascontiguousarray(synthetic.PIXEL_MEANS[::-1])


def close(got, ref, tol=1e-4):
    scale = max(1e-30, float(np.abs(ref).max()))
    err = float(np.abs(np.asarray(got, np.float64) - ref).max()) / scale
    assert err < tol, err


# (B, Cin, H, W, Cout, k, s, p): every kernel size / stride of the encoder, ragged channel counts, Ho*Wo % 4 == 0
WG_CASES = [(2, 8, 32, 40, 64, 7, 2, 3), (1, 64, 24, 32, 128, 5, 2, 2), (3, 24, 16, 20, 136, 3, 1, 1), (2, 40, 16, 24, 70, 3, 2, 1),
            (1, 5, 12, 12, 3, 3, 1, 1), (2, 256, 8, 10, 256, 3, 1, 1)]


@pytest.mark.parametrize("case", WG_CASES)
def test_conv_backward_kernels_match_oracle(ctx, case):
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    y = rng.standard_normal((B, cout, ho, wo)).astype(np.float32)          # the saved layer output (sign pattern)
    dy = rng.standard_normal((B, cout, ho, wo)).astype(np.float32)
    dz_ref = onet.lrelu_backward(dy, y, 0.1)
    dx_ref, dw_ref, db_ref = onet.conv2d_backward(x, w, dz_ref, s, p)
    h = ctx.handle
    dz = ctx.empty(dy.shape)
    lib.deepim_lrelu_backward(h, dz, ctx.array(dy), ctx.array(y), cf(0.1), dy.size)
    np.testing.assert_array_equal(dz.asnumpy(), dz_ref)
    db, dw = ctx.empty((cout,)), ctx.empty(w.shape)
    lib.deepim_bias_grad(h, db, dz, B, cout, ho * wo)
    lib.deepim_conv2d_wgrad(h, dw, ctx.array(x), dz, B, cin, H, W, cout, k, k, s, p)
    close(db.asnumpy(), db_ref)
    close(dw.asnumpy(), dw_ref)
    dw2 = ctx.empty(w.shape)                                                 # deterministic: bit-identical on a second call
    lib.deepim_conv2d_wgrad(h, dw2, ctx.array(x), dz, B, cin, H, W, cout, k, k, s, p)
    np.testing.assert_array_equal(dw2.asnumpy(), dw.asnumpy())
    # data gradient = forward kernel on the (dilated) dz with transposed + flipped weights
    wt = ctx.empty((cin, cout, k, k))
    lib.deepim_conv_flip_weights(h, wt, ctx.array(w), cout, cin, k, k)
    np.testing.assert_array_equal(wt.asnumpy(), np.ascontiguousarray(w.transpose(1, 0, 2, 3)[:, :, ::-1, ::-1]))
    pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cin, cout, k, k) // 4,))
    lib.deepim_conv_pack_weights(h, pk, wt, cin, cout, k, k)
    g, gh, gw = dz, ho, wo
    if s > 1:
        gh, gw = H - k + 1 + 2 * p, W - k + 1 + 2 * p
        g = ctx.empty((B, cout, gh, gw))
        lib.deepim_dilate2d(h, g, dz, B * cout, ho, wo, gh, gw, s)
        gd = g.asnumpy()
        assert np.array_equal(gd[:, :, ::s, ::s][:, :, :ho, :wo], dz_ref) and np.count_nonzero(gd) == np.count_nonzero(dz_ref)
    dx = ctx.empty(x.shape)
    lib.deepim_conv2d_forward(h, dx, g, pk, None, B, cout, gh, gw, cin, k, k, 1, k - 1 - p, cf(1.0), 0, 0)
    close(dx.asnumpy(), dx_ref)
    # the same packed weights without the flip pass: the transposed + flipped view read inside the pack kernels
    pk2 = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cin, cout, k, k) // 4,))
    lib.deepim_memset(h, pk2, 0, pk2.nbytes)
    lib.deepim_conv_pack_dgrad(h, pk2, ctx.array(w), cout, cin, k, k, 0, 0, 1, k, k, 3)
    half = lib.load().deepim_conv_packed_size(cin, cout, k, k) // 4 // 3
    np.testing.assert_array_equal(pk2.asnumpy()[:2 * half], pk.asnumpy()[:2 * half])


# (B, Cin, H, W, Cout, k, p): stride-2 layers incl. odd frame sizes (conv6: 15x20 -> 8x10) and a 7x7 kernel
S2_CASES = [(2, 64, 24, 32, 128, 5, 2), (2, 40, 16, 24, 70, 3, 1), (1, 512, 15, 20, 1024, 3, 1), (2, 6, 13, 17, 10, 3, 1),
            (1, 8, 20, 28, 16, 7, 3), (2, 128, 30, 40, 256, 3, 1), (4, 512, 15, 20, 1024, 3, 1), (3, 130, 9, 11, 258, 5, 2)]


@pytest.mark.parametrize("case", S2_CASES)
def test_stride2_dgrad_by_parity_classes_matches_oracle(ctx, case):
    """The data gradient of a stride-2 convolution as four stride-1 convolutions of the un-dilated dz (sub-kernel of each
    output parity class, full convolution, window interleaved into dx) — against the oracle's conv2d_backward and against
    the round-2 zero-dilated formulation (same sums in another order)."""
    B, cin, H, W, cout, k, p = case
    rng = np.random.default_rng(sum(case) + 5)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    ho, wo = (H + 2 * p - k) // 2 + 1, (W + 2 * p - k) // 2 + 1
    dz_h = rng.standard_normal((B, cout, ho, wo)).astype(np.float32)
    dx_ref, _, _ = onet.conv2d_backward(x, w, dz_h, 2, p)
    h = ctx.handle
    dz, wd = ctx.array(dz_h), ctx.array(w)
    dx = ctx.array(np.full(x.shape, 7.0, np.float32))          # every element must be written by exactly one class
    wt = ctx.empty((cin * cout * k * k,))
    pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cin, cout, k, k) // 4,))
    taps = 0
    for py in range(2):
        for px in range(2):
            ky0, kx0 = (py + p) % 2, (px + p) % 2
            nky, nkx = (k - ky0 + 1) // 2, (k - kx0 + 1) // 2
            taps += nky * nkx
            cy0, cx0 = (py + p - ky0) // 2, (px + p - kx0) // 2
            P = max(nky, nkx) - 1
            hf, wf = ho + 2 * P - nky + 1, wo + 2 * P - nkx + 1
            lib.deepim_conv_subkernel_flip(h, wt, wd, cout, cin, k, k, ky0, kx0, nky, nkx)
            sub = wt.asnumpy()[:cin * cout * nky * nkx].reshape(cin, cout, nky, nkx)
            np.testing.assert_array_equal(sub, np.ascontiguousarray(w[:, :, ky0::2, kx0::2].transpose(1, 0, 2, 3)[:, :, ::-1, ::-1]))
            lib.deepim_conv_pack_weights_ex(h, pk, wt, cin, cout, nky, nkx, 3)
            cls = ctx.empty((B, cin, hf, wf))
            lib.deepim_conv2d_forward(h, cls, dz, pk, None, B, cout, ho, wo, cin, nky, nkx, 1, P, cf(1.0), 0, 0)
            lib.deepim_interleave2d(h, dx, cls, B * cin, hf, wf, cy0 + P - (nky - 1), cx0 + P - (nkx - 1), H, W, py, px)
    assert taps == k * k                                       # the four classes partition the kernel: the ideal multiply-adds
    got = dx.asnumpy()
    close(got, dx_ref)
    # the same four convolutions storing their window straight onto dx (what the training graph runs): identical values
    dx2 = ctx.array(np.full(x.shape, 7.0, np.float32))
    for py in range(2):
        for px in range(2):
            ky0, kx0 = (py + p) % 2, (px + p) % 2
            nky, nkx = (k - ky0 + 1) // 2, (k - kx0 + 1) // 2
            cy0, cx0 = (py + p - ky0) // 2, (px + p - kx0) // 2
            P = max(nky, nkx) - 1
            order = lib.load().deepim_conv_weight_order(h, B, cout, ho, wo, cin, nky, nkx, 1, P)
            lib.deepim_conv_pack_dgrad(h, pk, wd, cout, cin, k, k, ky0, kx0, 2, nky, nkx, order)     # sub-kernel view inside the pack
            lib.deepim_conv2d_forward_remap(h, dx2, dz, pk, B, cout, ho, wo, cin, nky, nkx, P, cy0 + P - (nky - 1),
                                            cx0 + P - (nkx - 1), H, W, py, px)
    np.testing.assert_array_equal(dx2.asnumpy(), got)
    # rows / columns no output pixel reaches (odd frame, pad) get exact zeros, like the oracle
    assert np.abs(got[dx_ref == 0]).max(initial=0.0) == 0.0
    # the one-call form: class by class it is the loop above; grouped (default) the four classes share one pack / convolution /
    # second-pass launch with a joint split-K plan — other slice boundaries, so equal to the oracle, not bit-equal to the loop
    ws = DeviceArray(ctx, (lib.load().deepim_conv_dgrad_s2_packed_size(cout, cin, k, p) // 4,))
    try:
        lib.deepim_set_option(h, b"dgrad_group", 0)
        dx3 = ctx.array(np.full(x.shape, 7.0, np.float32))
        lib.deepim_conv2d_dgrad_s2(h, dx3, dz, wd, ws, B, cin, H, W, cout, k, p)
        np.testing.assert_array_equal(dx3.asnumpy(), got)
    finally:
        lib.deepim_set_option(h, b"dgrad_group", 1)
    res = []
    for _ in range(2):
        dx4 = ctx.array(np.full(x.shape, 7.0, np.float32))
        lib.deepim_conv2d_dgrad_s2(h, dx4, dz, wd, ws, B, cin, H, W, cout, k, p)
        res.append(dx4.asnumpy())
    close(res[0], dx_ref)
    np.testing.assert_array_equal(res[0], res[1])          # deterministic
    assert np.abs(res[0][dx_ref == 0]).max(initial=0.0) == 0.0


DG_CASES = [(2, 256, 8, 10, 256, 3, 1, 1), (3, 24, 16, 20, 136, 3, 1, 1), (1, 5, 12, 12, 3, 3, 1, 1), (2, 128, 30, 40, 130, 3, 1, 1),
            (2, 64, 24, 32, 128, 5, 2, 2), (4, 512, 15, 20, 1024, 3, 2, 1), (2, 6, 13, 17, 10, 3, 2, 1), (1, 8, 20, 28, 17, 7, 2, 3),
            (2, 128, 30, 40, 256, 3, 2, 1)]


@pytest.mark.parametrize("case", DG_CASES)
@pytest.mark.parametrize("with_add", [False, True])
def test_dgrad_with_activation_gradient_epilogue(ctx, case, with_add):
    """deepim_conv2d_dgrad: the plain data gradient against the oracle; with act_y [and add] the result is bit-identical to the
    plain one followed by (dx + add) · lrelu'(act_y) — whether the epilogue rode in the convolution's stores (register-fed
    kernels, grouped stride-2 classes, split-K second passes) or ran as a pass of its own (other kernel families)."""
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(sum(case) + 21)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dz_h = rng.standard_normal((B, cout, ho, wo)).astype(np.float32)
    dx_ref, _, _ = onet.conv2d_backward(x, w, dz_h, s, p)
    h = ctx.handle
    dz, wd = ctx.array(dz_h), ctx.array(w)
    ws = DeviceArray(ctx, (lib.load().deepim_conv_dgrad_packed_size(cout, cin, k, s, p) // 4,))
    plain = ctx.array(np.full(x.shape, 7.0, np.float32))
    lib.deepim_conv2d_dgrad(h, plain, dz, wd, ws, B, cin, H, W, cout, k, s, p, None, None, cf(0.1))
    close(plain.asnumpy(), dx_ref)
    y = rng.standard_normal(x.shape).astype(np.float32)
    add = rng.standard_normal(x.shape).astype(np.float32) if with_add else None
    got = ctx.array(np.full(x.shape, 7.0, np.float32))
    lib.deepim_conv2d_dgrad(h, got, dz, wd, ws, B, cin, H, W, cout, k, s, p, ctx.array(y), ctx.array(add) if with_add else None, cf(0.1))
    v = plain.asnumpy() + add if with_add else plain.asnumpy()
    np.testing.assert_array_equal(got.asnumpy(), np.where(y > 0, v, v * np.float32(0.1)).astype(np.float32))


@pytest.mark.parametrize("case", WG_CASES + [(4, 64, 60, 80, 128, 5, 2, 2), (2, 512, 15, 20, 1024, 3, 2, 1),
                                  # few filters: the stream kernel (heads: 3x3 s1; flow upsamplers' role-swapped k4 s2)
                                  (4, 770, 30, 40, 2, 3, 1, 1), (2, 6, 30, 40, 1, 3, 1, 1), (2, 2, 32, 42, 2, 4, 2, 0),
                                  (3, 9, 10, 12, 4, 3, 1, 1), (2, 1026, 15, 20, 2, 3, 1, 1)])
def test_wgrad_lds_kernel_vs_register_fed_kernel(ctx, case):
    """The LDS-staged weight-gradient kernel (default) and the round-2 register-fed one: both within 1e-4 of the float64
    oracle, and within fp32 re-association distance of each other; both deterministic."""
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(sum(case) + 9)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dz_h = rng.standard_normal((B, cout, ho, wo)).astype(np.float32)
    w0 = np.zeros((cout, cin, k, k), np.float32)
    _, dw_ref, _ = onet.conv2d_backward(x, w0, dz_h, s, p)
    h = ctx.handle
    xd, dz = ctx.array(x), ctx.array(dz_h)
    res = {}
    try:
        for mode in (1, 0):
            lib.deepim_set_option(h, b"wgrad_lds", mode)
            dw = ctx.empty(w0.shape)
            lib.deepim_conv2d_wgrad(h, dw, xd, dz, B, cin, H, W, cout, k, k, s, p)
            dw2 = ctx.empty(w0.shape)
            lib.deepim_conv2d_wgrad(h, dw2, xd, dz, B, cin, H, W, cout, k, k, s, p)
            res[mode] = dw.asnumpy()
            np.testing.assert_array_equal(dw2.asnumpy(), res[mode])
            close(res[mode], dw_ref)
    finally:
        lib.deepim_set_option(h, b"wgrad_lds", 1)
    close(res[1], res[0].astype(np.float64), 1e-5)


@pytest.mark.parametrize("case", [c for c in WG_CASES if c[1] % 8 == 0 and c[4] > 4] +
                         [(4, 64, 60, 80, 128, 5, 2, 2), (2, 512, 15, 20, 1024, 3, 2, 1), (1, 8, 480, 640, 64, 7, 2, 3), (2, 256, 32, 42, 1026, 4, 2, 0)])
def test_tap_major_wgrad_is_the_natural_one_permuted(ctx, case):
    """deepim_conv2d_wgrad_tm: (Cout, kh*kw, Cin) layout, bit-identical to deepim_conv2d_wgrad after the permutation (same pixel
    chunks, same slices, same order — only the K rows move); deepim_weight_grad_to_natural and the SGD kernel's layout word
    both apply that permutation."""
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(sum(case) + 33)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dz_h = rng.standard_normal((B, cout, ho, wo)).astype(np.float32)
    h = ctx.handle
    xd, dz = ctx.array(x), ctx.array(dz_h)
    nat = ctx.empty((cout, cin, k, k))
    lib.deepim_conv2d_wgrad(h, nat, xd, dz, B, cin, H, W, cout, k, k, s, p)
    tm = ctx.empty((cout, k * k, cin))
    lib.deepim_conv2d_wgrad_tm(h, tm, xd, dz, B, cin, H, W, cout, k, k, s, p)
    want = nat.asnumpy()
    np.testing.assert_array_equal(tm.asnumpy(), want.reshape(cout, cin, k * k).transpose(0, 2, 1))
    back = ctx.empty((cout, cin, k, k))
    lib.deepim_weight_grad_to_natural(h, back, tm, cout, cin, k * k)
    np.testing.assert_array_equal(back.asnumpy(), want)
    # SGD on the tap-major gradient in place == SGD on the natural one
    w0 = rng.standard_normal(want.shape).astype(np.float32)
    res = []
    for g, layout in ((nat, 0), (tm, cin | ((k * k) << 32))):
        w, m = ctx.array(w0), ctx.zeros(want.shape)
        tab = ctx.empty((1, 6), np.uint64)
        tab.copyfrom(np.array([[w.ptr, m.ptr, g.ptr, w.size, int(np.array([5e-4], np.float32).view(np.uint32)[0]), layout]], dtype=np.uint64))
        lib.deepim_sgd_mom_update_multi(h, tab, 1, (w.size + 1023) // 1024, cf(1e-3), cf(0.975), cf(1.0), cf(0.0))
        res.append((w.asnumpy(), m.asnumpy()))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])


@pytest.mark.parametrize("case", [(4, 770, 30, 40, 2, 3, 1, 1), (2, 6, 30, 40, 1, 3, 1, 1), (3, 9, 10, 12, 4, 3, 1, 1), (2, 24, 16, 20, 136, 3, 1, 1)])
def test_wgrad_with_bias_in_one_call(ctx, case):
    """deepim_conv2d_wgrad_bias = deepim_conv2d_wgrad + the bias gradient (in the same launch for the few-filter layers)."""
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(sum(case) + 3)
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dz_h = rng.standard_normal((B, cout, ho, wo)).astype(np.float32)
    h = ctx.handle
    xd, dz = ctx.array(x), ctx.array(dz_h)
    dw_ref = ctx.empty((cout, cin, k, k))
    lib.deepim_conv2d_wgrad(h, dw_ref, xd, dz, B, cin, H, W, cout, k, k, s, p)
    dw, db = ctx.empty((cout, cin, k, k)), ctx.empty((cout,))
    lib.deepim_conv2d_wgrad_bias(h, dw, db, xd, dz, B, cin, H, W, cout, k, k, s, p)
    np.testing.assert_array_equal(dw.asnumpy(), dw_ref.asnumpy())
    close(db.asnumpy(), dz_h.astype(np.float64).sum(axis=(0, 2, 3)), 1e-5)


@pytest.mark.parametrize("shape", [(4, 81920, 256), (3, 256, 256), (5, 256, 7)])
def test_fc_backward_and_sgd(ctx, shape):
    B, I, O = shape
    rng = np.random.default_rng(B + O)
    x = rng.standard_normal((B, I)).astype(np.float32)
    w = (rng.standard_normal((O, I)) / np.sqrt(I)).astype(np.float32)
    dy = rng.standard_normal((B, O)).astype(np.float32)
    dx_ref, dw_ref, db_ref = onet.fc_backward(x, w, dy)
    dx, dw, db = ctx.empty(x.shape), ctx.empty(w.shape), ctx.empty((O,))
    lib.deepim_fc_backward(ctx.handle, dx, dw, db, ctx.array(dy), ctx.array(x), ctx.array(w), B, I, O)
    close(dx.asnumpy(), dx_ref); close(dw.asnumpy(), dw_ref); close(db.asnumpy(), db_ref)
    mom = (rng.standard_normal(w.shape) * 1e-3).astype(np.float32)
    for clip in (None, 0.01):
        wd_, md_ = ctx.array(w), ctx.array(mom)
        lib.deepim_sgd_mom_update(ctx.handle, wd_, md_, dw, cf(1e-4), cf(5e-4), cf(0.975), cf(0.5), cf(clip or 0.0), w.size)
        w_ref, m_ref = onet.sgd_mom_update(w, mom, dw.asnumpy(), 1e-4, 5e-4, 0.975, 0.5, clip)
        np.testing.assert_allclose(wd_.asnumpy(), w_ref, rtol=1e-6, atol=1e-9)
        np.testing.assert_allclose(md_.asnumpy(), m_ref, rtol=1e-5, atol=1e-10)


@pytest.mark.parametrize("shape", [(4, 64, 240, 320), (2, 1024, 8, 10), (3, 5, 7, 9), (1, 130, 30, 40)])
@pytest.mark.parametrize("with_add", [False, True])
def test_lrelu_bias_backward_equals_the_separate_passes(ctx, shape, with_add):
    """The fused walk (skip add + LeakyReLU gradient + bias-gradient first pass): dz bit-identical to axpy → lrelu_backward,
    db within float64-accumulation distance of deepim_bias_grad on that dz (and of the numpy sum); in place over dy."""
    B, C, H, W = shape
    rng = np.random.default_rng(sum(shape))
    dy = rng.standard_normal(shape).astype(np.float32)
    y = rng.standard_normal(shape).astype(np.float32)
    add = rng.standard_normal(shape).astype(np.float32) if with_add else None
    h = ctx.handle
    ref = ctx.array(dy)
    if with_add:
        lib.deepim_axpy(h, ref, ctx.array(add), cf(1.0), ref.size)
    lib.deepim_lrelu_backward(h, ref, ref, ctx.array(y), cf(0.1), ref.size)
    db_ref = ctx.empty((C,))
    lib.deepim_bias_grad(h, db_ref, ref, B, C, H * W)
    g, db = ctx.array(dy), ctx.empty((C,))
    lib.deepim_lrelu_bias_backward(h, g, db, g, ctx.array(add) if with_add else None, ctx.array(y), cf(0.1), B, C, H * W)
    np.testing.assert_array_equal(g.asnumpy(), ref.asnumpy())
    host = ref.asnumpy().astype(np.float64).sum(axis=(0, 2, 3))
    np.testing.assert_allclose(db.asnumpy(), host, rtol=2e-6, atol=1e-6 * np.sqrt(B * H * W))
    np.testing.assert_allclose(db.asnumpy(), db_ref.asnumpy(), rtol=2e-6, atol=1e-6 * np.sqrt(B * H * W))


def test_backward_entries_on_an_empty_batch(ctx):
    """B = 0 (a rank whose shard is empty): the new entries return success, touch nothing they should not, and a bias gradient
    over no elements is zero."""
    h = ctx.handle
    db = ctx.array(np.full((6,), 7.0, np.float32))
    x = ctx.array(np.zeros((1, 6, 4, 4), np.float32))
    lib.deepim_lrelu_bias_backward(h, x, db, x, None, x, cf(0.1), 0, 6, 16)
    np.testing.assert_array_equal(db.asnumpy(), np.zeros(6, np.float32))
    w = ctx.array(np.ones((8, 8, 3, 3), np.float32))
    ws = DeviceArray(ctx, (lib.load().deepim_conv_dgrad_packed_size(8, 8, 3, 2, 1) // 4,))
    keep = ctx.array(np.full((1, 8, 4, 4), 3.0, np.float32))
    for stride in (1, 2):
        lib.deepim_conv2d_dgrad(h, keep, x, w, ws, 0, 8, 4, 4, 8, 3, stride, 1, None, None, cf(0.1))
    # a weight / bias gradient over no samples is zero, not whatever the buffer held (update() would apply it): ADVICE r3
    dw = ctx.array(np.full((8, 9, 8), 5.0, np.float32))
    lib.deepim_conv2d_wgrad_tm(h, dw, keep, keep, 0, 8, 4, 4, 8, 3, 3, 1, 1)
    dw2, db2 = ctx.array(np.full((8, 8, 3, 3), 5.0, np.float32)), ctx.array(np.full((8,), 5.0, np.float32))
    lib.deepim_conv2d_wgrad_bias(h, dw2, db2, keep, keep, 0, 8, 4, 4, 8, 3, 3, 1, 1)
    ctx.sync()
    assert (keep.asnumpy() == 3.0).all()
    assert not dw.asnumpy().any() and not dw2.asnumpy().any() and not db2.asnumpy().any()
    # deepim_bias_grad with no elements per channel (hw = 0 used to divide by zero) or no samples
    for B_, hw_ in ((2, 0), (0, 16)):
        db3 = ctx.array(np.full((6,), 7.0, np.float32))
        lib.deepim_bias_grad(h, db3, x, B_, 6, hw_)
        np.testing.assert_array_equal(db3.asnumpy(), np.zeros(6, np.float32))


def test_sgd_multi_is_bit_identical_to_per_tensor_updates(ctx):
    """deepim_sgd_mom_update_multi: one launch over a table of parameters (ragged sizes, weight decay per row) — the same bits as
    one deepim_sgd_mom_update per tensor."""
    rng = np.random.default_rng(5)
    sizes = [1, 255, 256, 257, 64 * 8 * 49, 7, 1024 * 3 + 5]
    h = ctx.handle
    W = [rng.standard_normal(n).astype(np.float32) for n in sizes]
    M = [(rng.standard_normal(n) * 1e-3).astype(np.float32) for n in sizes]
    G_ = [rng.standard_normal(n).astype(np.float32) for n in sizes]
    wds = [5e-4 if i % 2 == 0 else 0.0 for i in range(len(sizes))]
    for clip in (0.0, 0.3):
        ref_w, ref_m = [ctx.array(a) for a in W], [ctx.array(a) for a in M]
        g = [ctx.array(a) for a in G_]
        for i, n in enumerate(sizes):
            lib.deepim_sgd_mom_update(h, ref_w[i], ref_m[i], g[i], cf(1e-3), cf(wds[i]), cf(0.975), cf(0.5), cf(clip), n)
        w, m = [ctx.array(a) for a in W], [ctx.array(a) for a in M]
        rows, block = [], 0
        for i, n in enumerate(sizes):
            rows.append([w[i].ptr, m[i].ptr, g[i].ptr, n, int(np.array([wds[i]], np.float32).view(np.uint32)[0]) | (block << 32), 0])
            block += (n + 1023) // 1024
        tab = ctx.empty((len(rows), 6), np.uint64)
        tab.copyfrom(np.array(rows, dtype=np.uint64))
        lib.deepim_sgd_mom_update_multi(h, tab, len(rows), block, cf(1e-3), cf(0.975), cf(0.5), cf(clip))
        for i in range(len(sizes)):
            np.testing.assert_array_equal(w[i].asnumpy(), ref_w[i].asnumpy())
            np.testing.assert_array_equal(m[i].asnumpy(), ref_m[i].asnumpy())


def _train_setup(ctx, B, seed, pred_heads, input_mask=True, input_depth=False):
    d = synthetic.make_batch(B, seed=seed, n_frames=1)
    cfg = default_config()
    cfg.network.PRED_FLOW = cfg.network.PRED_MASK = pred_heads
    cfg.network.INPUT_MASK, cfg.network.INPUT_DEPTH = input_mask, input_depth
    net = deepIM_flownet().get_symbol(cfg, is_train=True)
    params = net.init_weights(cfg, seed=91)
    net.bind_train(ctx, B, params, num_points=3000)
    gt = (d["depth_gt_observed"] > 0).astype(np.float32)
    pco = np.stack([d["pose_tgt"][b][:, :3].astype(np.float64) @ d["point_cloud_model"][b].astype(np.float64) + d["pose_tgt"][b][:, 3:4]
                    for b in range(B)]).astype(np.float32)
    wts = np.ones((B, 3, 3000), np.float32)
    data_np = {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][0], "mask_observed": d["mask_observed"],
               "mask_rendered": d["mask_rendered"][0], "src_pose": d["src_pose"][0]}
    if input_depth:
        data_np.update(depth_observed=d["depth_gt_observed"], depth_rendered=d["depth_rendered"][0])
    label_np = {"mask_gt_observed": gt, "point_cloud_model": d["point_cloud_model"], "point_cloud_weights": wts,
                "point_cloud_observed": pco}
    if pred_heads:   # flow labels the way the data layer makes them (lib/pair_matching/data_pair.py:get_pair_flow, on the device)
        from mx_deepim_amd.lib.pair_matching import data_pair
        flow, fw = data_pair.get_pair_flow({"depth_rendered": ctx.array(d["depth_rendered"][0]),
                                            "depth_gt_observed": ctx.array(d["depth_gt_observed"]),
                                            "pose_rendered": ctx.array(d["src_pose"][0]), "pose_observed": ctx.array(d["pose_tgt"])}, cfg)
        label_np["flow"], label_np["flow_weights"] = flow.asnumpy(), fw.asnumpy()
        assert np.count_nonzero(label_np["flow_weights"]) > 1000
    return d, cfg, net, params, data_np, label_np


def test_training_iteration_of_the_pose_branch_matches_oracle(ctx):
    """One training-style iteration (B = 1, 480x640): zoom from the gt mask → encoder → fc → rot/trans → Transform3D →
    point-matching loss, backward through everything, SGD step (module.py:1131-1137 order)."""
    B = 1
    d, cfg, net, params, data_np, label_np = _train_setup(ctx, B, 910, False)
    data = {k: ctx.array(v) for k, v in data_np.items()}
    label = {k: ctx.array(v) for k, v in label_np.items()}
    # LeakyReLU makes the gradient discontinuous where an activation crosses zero: run the forward convs as single canonical
    # fmaf chains (bit-identical to the oracle's) so that both sides differentiate at exactly the same activations
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    try:
        loss = net.forward_train(data, label).asnumpy()[0]
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    grads = net.backward()
    t = cfg.train_iter
    ref_loss, g_ref, fwd = opipe.train_pose_iteration(params, data_np, label_np, d["K"], MEANS_REV, cfg.dataset.trans_means,
                                                      cfg.dataset.trans_stds, cfg.network.ROT_COORD, t.LW_PM, t.NUM_3D_SAMPLE,
                                                      cfg.dataset.NORMALIZE_3D_POINT, t.SE3_PM_LOSS_TYPE, t.SE3_PM_SL1_SCALAR)
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), fwd["net_input"])
    np.testing.assert_array_equal(net.act["conv6_1"].asnumpy(), fwd["conv6_1"])
    close(net.act["points_est"].asnumpy(), fwd["points_est"])
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss)
    assert set(grads) == set(g_ref)
    for name in sorted(g_ref):
        assert np.abs(g_ref[name]).max() > 0, name
        close(grads[name].asnumpy(), g_ref[name], 2e-4)
    # SGD step, then the forward really uses the updated (re-packed) weights: the loss changes and stays finite
    before = {k: v.asnumpy() for k, v in net.params.items() if k in ("fc6_weight", "flow_conv1_weight")}
    net.update(lr=1e-2, wd=cfg.TRAIN.wd, momentum=cfg.TRAIN.momentum)
    for k, v in before.items():
        w_ref, _ = onet.sgd_mom_update(v, np.zeros_like(v), g_ref[k], 1e-2, cfg.TRAIN.wd, cfg.TRAIN.momentum)
        np.testing.assert_allclose(net.params[k].asnumpy(), w_ref, rtol=1e-4, atol=1e-7)
    loss2 = net.forward_train(data, label).asnumpy()[0]
    assert np.isfinite(loss2) and loss2 != loss


@pytest.mark.parametrize("input_mask,input_depth", [(False, False), (True, True)])
def test_training_iteration_with_6_and_10_input_channels(ctx, input_mask, input_depth):
    """INPUT_MASK = False (C_in = 6) and INPUT_DEPTH + INPUT_MASK (C_in = 10): conv1's weight gradient cannot take the tap-major
    LDS entry (C_in % 8 != 0) — bind_train registers the natural buffer for it and backward() calls deepim_conv2d_wgrad (ADVICE r3:
    this raised). Gradients against the oracle, then an SGD step that really moves conv1."""
    B = 1
    d, cfg, net, params, data_np, label_np = _train_setup(ctx, B, 913, False, input_mask, input_depth)
    assert net.cin == (10 if input_depth else 6) and "flow_conv1_weight" not in net.grad.tm and "conv2_weight" in net.grad.tm
    data = {k: ctx.array(v) for k, v in data_np.items()}
    label = {k: ctx.array(v) for k, v in label_np.items()}
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    try:
        loss = net.forward_train(data, label).asnumpy()[0]
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    grads = net.backward()
    t = cfg.train_iter
    odata = dict(data_np)
    if not input_mask:
        odata["mask_observed"] = odata["mask_rendered"] = None      # oracle/zoom.py: ZoomImage computes the factor
    ref_loss, g_ref, fwd = opipe.train_pose_iteration(params, odata, label_np, d["K"], MEANS_REV, cfg.dataset.trans_means,
                                                      cfg.dataset.trans_stds, cfg.network.ROT_COORD, t.LW_PM, t.NUM_3D_SAMPLE,
                                                      cfg.dataset.NORMALIZE_3D_POINT, t.SE3_PM_LOSS_TYPE, t.SE3_PM_SL1_SCALAR)
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), fwd["net_input"])
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss)
    for name in sorted(g_ref):
        close(grads[name].asnumpy(), g_ref[name], 2e-4)
    before = net.params["flow_conv1_weight"].asnumpy()
    net.update(lr=1e-2, wd=cfg.TRAIN.wd, momentum=cfg.TRAIN.momentum)
    w_ref, _ = onet.sgd_mom_update(before, np.zeros_like(before), g_ref["flow_conv1_weight"], 1e-2, cfg.TRAIN.wd, cfg.TRAIN.momentum)
    np.testing.assert_allclose(net.params["flow_conv1_weight"].asnumpy(), w_ref, rtol=1e-4, atol=1e-7)


def test_rebind_and_register_fed_wgrad_switch(ctx):
    """(a) bind_train twice on one net object: update() must write the NEW parameters (its pointer table is rebuilt, ADVICE r3 —
    the cached table pointed at the freed buffers of the first bind). (b) wgrad_lds = 0, the documented A/B switch: the training
    step runs on the register-fed kernels with natural-layout gradients and agrees with the default path."""
    B = 1
    d, cfg, net, params, data_np, label_np = _train_setup(ctx, B, 915, False)
    data = {k: ctx.array(v) for k, v in data_np.items()}
    label = {k: ctx.array(v) for k, v in label_np.items()}
    net.forward_train(data, label)
    g1 = {k: v.asnumpy() for k, v in net.backward().items()}
    net.update(lr=1e-2, wd=cfg.TRAIN.wd, momentum=cfg.TRAIN.momentum)
    w_after_1 = net.params["conv3_weight"].asnumpy()
    assert not np.array_equal(w_after_1, params["conv3_weight"])
    lib.deepim_set_option(ctx.handle, b"wgrad_lds", 0)
    try:
        net.bind_train(ctx, B, params, num_points=3000)        # second bind: new params / mom / grad buffers
        assert not net.grad.tm
        np.testing.assert_array_equal(net.params["conv3_weight"].asnumpy(), params["conv3_weight"])
        net.forward_train(data, label)
        g2 = {k: v.asnumpy() for k, v in net.backward().items()}
        for k in sorted(g1):
            close(g2[k], g1[k].astype(np.float64), 2e-5)
        net.update(lr=1e-2, wd=cfg.TRAIN.wd, momentum=cfg.TRAIN.momentum)
        np.testing.assert_allclose(net.params["conv3_weight"].asnumpy(), w_after_1, rtol=1e-4, atol=1e-7)
    finally:
        lib.deepim_set_option(ctx.handle, b"wgrad_lds", 1)


def test_backward_on_two_streams_is_bit_identical(ctx):
    """backward() with the weight gradients on a second context / stream (deepim_stream_wait orders the two; off by default:
    measured no faster) gives the same bits as the one-stream order, run after run — with the decoder and both heads."""
    from mx_deepim_amd.runtime import Context
    B = 2
    d, cfg, net, params, data_np, label_np = _train_setup(ctx, B, 77, True)
    data = {k: ctx.array(v) for k, v in data_np.items()}
    label = {k: ctx.array(v) for k, v in label_np.items()}
    net.forward_train(data, label)
    one = {k: v.asnumpy() for k, v in net.backward().items()}
    net.side = Context(ctx.device_id)          # what bind_train does under net.two_streams = True
    for _ in range(3):
        two = net.backward()
        ctx.sync()
        for k in sorted(one):
            np.testing.assert_array_equal(two[k].asnumpy(), one[k], err_msg=k)
    # the primitive on its own: B queues behind A without the host waiting in between
    a, b = ctx, net.side
    x = a.zeros((1 << 22,))
    for _ in range(20):
        lib.deepim_axpy(a.handle, x, a.array(np.ones(1 << 22, np.float32)), cf(1.0), x.size)
    lib.deepim_stream_wait(b.handle, a.handle)
    y = DeviceArray(b, x.shape)
    lib.deepim_d2d(b.handle, y, x, x.nbytes)
    b.sync()
    np.testing.assert_array_equal(y.asnumpy(), np.full(1 << 22, 20.0, np.float32))


DEC_CASES = [(2, 70, 6, 8, 24, 13, 17), (1, 2, 8, 10, 2, 15, 20), (1, 1026, 15, 20, 256, 30, 40)]


@pytest.mark.parametrize("case", DEC_CASES)
def test_deconv_crop_backward_matches_oracle(ctx, case):
    """Deconvolution k4 s2 + Crop(1,1) backward as composed in deepIM_flownet._deconv_backward: un-crop (scatter2d), conv wgrad
    with input and output swapped, stride-2 forward conv for the data gradient."""
    B, cin, H, W, cout, ho, wo = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cin, cout, 4, 4)) / np.sqrt(cin * 4)).astype(np.float32)
    dy = rng.standard_normal((B, cout, ho, wo)).astype(np.float32)
    dx_ref, dw_ref, db_ref = onet.deconv4x4s2_crop_backward(x, w, dy, (1, 1))
    h = ctx.handle
    hf, wf = 2 * H + 2, 2 * W + 2
    full = ctx.empty((B, cout, hf, wf))
    lib.deepim_scatter2d(h, full, ctx.array(dy), B * cout, ho, wo, hf, wf, 1, 1, 1)
    f = full.asnumpy()
    np.testing.assert_array_equal(f[:, :, 1:1 + ho, 1:1 + wo], dy)
    assert np.count_nonzero(f) == np.count_nonzero(dy)
    db, dw, dx = ctx.empty((cout,)), ctx.empty(w.shape), ctx.empty(x.shape)
    lib.deepim_bias_grad(h, db, ctx.array(dy), B, cout, ho * wo)
    lib.deepim_conv2d_wgrad(h, dw, full, ctx.array(x), B, cout, hf, wf, cin, 4, 4, 2, 0)
    pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cin, cout, 4, 4) // 4,))
    lib.deepim_conv_pack_weights(h, pk, ctx.array(w), cin, cout, 4, 4)
    lib.deepim_conv2d_forward(h, dx, full, pk, None, B, cout, hf, wf, cin, 4, 4, 2, 0, cf(1.0), 0, 0)
    close(db.asnumpy(), db_ref); close(dw.asnumpy(), dw_ref); close(dx.asnumpy(), dx_ref)
    # the one-walk front of it (what the training graph runs): slice of a concat gradient x lrelu'(slice of the saved concat) →
    # un-cropped frame + bias gradient; bit-identical frame, same bias sum
    ctotal, coff = cout + 5, 3
    dcat = rng.standard_normal((B, ctotal, ho, wo)).astype(np.float32)
    ycat = rng.standard_normal((B, ctotal, ho, wo)).astype(np.float32)
    for use_y in (True, False):
        sl = dcat[:, coff:coff + cout]
        ref = np.where(ycat[:, coff:coff + cout] > 0, sl, sl * np.float32(0.1)).astype(np.float32) if use_y else sl
        out, db2 = ctx.array(np.full((B, cout, hf, wf), 7.0, np.float32)), ctx.empty((cout,))
        lib.deepim_slice_lrelu_bias_scatter(h, out, db2, ctx.array(dcat), ctx.array(ycat) if use_y else None, B, ctotal, coff, cout,
                                            ho, wo, hf, wf, 1, 1, cf(0.1))
        want = np.zeros((B, cout, hf, wf), np.float32)
        want[:, :, 1:1 + ho, 1:1 + wo] = ref
        np.testing.assert_array_equal(out.asnumpy(), want)
        close(db2.asnumpy(), ref.astype(np.float64).sum(axis=(0, 2, 3)), 1e-6)


@pytest.mark.parametrize("shape", [(2, 2, 30, 40, 480, 640), (1, 1, 30, 40, 480, 640), (3, 2, 5, 7, 70, 100)])
def test_upsample16_backward_and_extract_channels(ctx, shape):
    B, C, H, W, Ho, Wo = shape
    rng = np.random.default_rng(Ho + C)
    w = np.stack([deepIM_flownet._init_bilinear((1, 1, 32, 32))[0] for _ in range(C)]).astype(np.float32)
    w = (w * rng.uniform(0.5, 1.5, w.shape)).astype(np.float32)           # not separable: every tap distinct
    dy = rng.standard_normal((B, C, Ho, Wo)).astype(np.float32)
    ref = onet.upsample16_crop_backward(dy, w, H, W, (8, 8), 1.5)
    out = ctx.empty((B, C, H, W))
    lib.deepim_upsample16_crop_backward(ctx.handle, out, ctx.array(dy), ctx.array(w), B, C, H, W, Ho, Wo, 8, 8, cf(1.5))
    close(out.asnumpy(), ref, 1e-5)
    src = rng.standard_normal((B, 11, H * W)).astype(np.float32)
    dst = ctx.empty((B, 4, H * W))
    lib.deepim_extract_channels(ctx.handle, dst, ctx.array(src), 11, 5, 4, B, H * W)
    np.testing.assert_array_equal(dst.asnumpy(), src[:, 5:9])


def test_training_iteration_with_flow_and_mask_heads_matches_oracle(ctx):
    """The full training graph (network.PRED_FLOW = PRED_MASK = True, the reference's default flow-net configuration): pose
    branch + refinement decoder + flow loss + mask loss, backward through all of it (skip connections into conv4_1 / conv5_1 /
    conv6_1 included), SGD step with the bilinear upsampling kernels held fixed."""
    B = 1
    d, cfg, net, params, data_np, label_np = _train_setup(ctx, B, 915, True)
    data = {k: ctx.array(v) for k, v in data_np.items()}
    label = {k: ctx.array(v) for k, v in label_np.items()}
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    try:
        loss = net.forward_train(data, label).asnumpy()[0]
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    grads = net.backward()
    t = cfg.train_iter
    ref_loss, g_ref, fwd = opipe.train_iteration(params, data_np, label_np, d["K"], MEANS_REV, cfg.dataset.trans_means,
                                                 cfg.dataset.trans_stds, cfg.network.ROT_COORD, t.LW_PM, t.NUM_3D_SAMPLE,
                                                 cfg.dataset.NORMALIZE_3D_POINT, t.SE3_PM_LOSS_TYPE, t.SE3_PM_SL1_SCALAR,
                                                 pred_flow=True, pred_mask=True, lw_flow=t.LW_FLOW, lw_mask=t.LW_MASK,
                                                 normalize_flow=cfg.dataset.NORMALIZE_FLOW)
    A = net.act
    np.testing.assert_array_equal(A["conv6_1"].asnumpy(), fwd["conv6_1"])
    np.testing.assert_array_equal(A["Concat3"].asnumpy(), fwd["Concat3"])
    np.testing.assert_array_equal(A["zoom_flow_gt"].asnumpy(), fwd["zoom_flow_gt"])
    np.testing.assert_array_equal(A["zoom_flow_weights"].asnumpy(), fwd["zoom_flow_weights"])
    np.testing.assert_array_equal(A["zoom_mask_gt_observed"].asnumpy(), fwd["zoom_mask_gt_observed"])
    close(A["zoom_flow_est"].asnumpy(), fwd["flow_est_crop"], 1e-5)
    close(A["flow_loss"].asnumpy(), fwd["flow_loss"], 1e-5)
    close(A["mask_prob"].asnumpy(), fwd["mask_prob"], 1e-5)
    assert abs(A["flow_loss_sum"].asnumpy()[0] - fwd["flow_loss_sum"]) <= 1e-4 * abs(fwd["flow_loss_sum"])
    assert abs(loss - ref_loss) <= 1e-4 * abs(ref_loss)
    close(net.ws["d_Concat3"].asnumpy(), fwd["d_Concat3"], 2e-4)
    close(net.ws["d_Concat2"].asnumpy(), fwd["d_Concat2"], 2e-4)
    assert set(grads) == set(g_ref)
    for name in sorted(g_ref):
        if name.endswith("upsampling_weight"):
            assert not grads[name].asnumpy().any()
            continue
        assert np.abs(g_ref[name]).max() > 0, name
        close(grads[name].asnumpy(), g_ref[name], 2e-4)
    # the heads really reach the encoder: its gradients differ from the pose-branch-only ones
    _, g_pose, _ = opipe.train_pose_iteration(params, data_np, label_np, d["K"], MEANS_REV, cfg.dataset.trans_means,
                                              cfg.dataset.trans_stds, cfg.network.ROT_COORD, t.LW_PM, t.NUM_3D_SAMPLE,
                                              cfg.dataset.NORMALIZE_3D_POINT, t.SE3_PM_LOSS_TYPE, t.SE3_PM_SL1_SCALAR)
    assert np.abs(g_pose["conv4_1_weight"] - g_ref["conv4_1_weight"]).max() > 1e-3 * np.abs(g_ref["conv4_1_weight"]).max()
    # SGD: weights move by the oracle's step, biases carry no weight decay, the bilinear kernels stay put, the decoder's packed
    # weights are refreshed (the forward changes)
    names = ("deconv4_weight", "Convolution3_weight", "deconv5_bias", "upsampling_weight", "mask_upsampling_weight")
    before = {k: net.params[k].asnumpy() for k in names}
    c3 = A["Concat3"].asnumpy()
    net.update(lr=1e-2, wd=cfg.TRAIN.wd, momentum=cfg.TRAIN.momentum)
    for k, v in before.items():
        if k.endswith("upsampling_weight"):
            np.testing.assert_array_equal(net.params[k].asnumpy(), v)
            continue
        w_ref, _ = onet.sgd_mom_update(v, np.zeros_like(v), g_ref[k], 1e-2, cfg.TRAIN.wd if k.endswith("_weight") else 0.0,
                                       cfg.TRAIN.momentum)
        np.testing.assert_allclose(net.params[k].asnumpy(), w_ref, rtol=1e-4, atol=1e-7)
    loss2 = net.forward_train(data, label).asnumpy()[0]
    assert np.isfinite(loss2) and loss2 != loss
    assert not np.array_equal(A["Concat3"].asnumpy(), c3)


def test_training_with_heads_reduces_all_three_losses(ctx):
    """A few SGD steps on one fixed batch (B = 2): the point-matching, flow and mask losses all go down."""
    B = 2
    d, cfg, net, params, data_np, label_np = _train_setup(ctx, B, 77, True)
    data = {k: ctx.array(v) for k, v in data_np.items()}
    label = {k: ctx.array(v) for k, v in label_np.items()}

    def losses():
        pm = net.forward_train(data, label).asnumpy()[0]
        p, y = net.act["mask_prob"].asnumpy().astype(np.float64), net.act["zoom_mask_gt_observed"].asnumpy()
        bce = float(-(y * np.log(p + 1e-12) + (1 - y) * np.log(1 - p + 1e-12)).mean())
        return float(pm), float(net.act["flow_loss_sum"].asnumpy()[0]), bce

    first = losses()
    for _ in range(6):
        net.backward()
        net.update(lr=2e-3, wd=cfg.TRAIN.wd, momentum=0.5)
        last = losses()
    assert all(np.isfinite(last))
    assert last[0] < first[0] and last[1] < first[1] and last[2] < first[2], (first, last)


def test_train_step_chains_iterations_through_the_batch_updater(ctx):
    """The reference's training step (module.py:1131-1137, TRAIN_ITER_SIZE iterations with batchUpdaterPyMulti.forward between
    them) composed on the device: `net.train_step`. B = 2, decoder + flow + mask heads, two chained iterations:
      * iteration 1 = forward_train / backward / update (checked against the oracle elsewhere in this file);
      * the update between them, against the oracle composition fed the GPU's own predictions: refined pose = RT_transform
        (<= 1e-6), rot / trans labels = calc_RT_delta (<= 1e-5), re-rendered depth bit-exact and colour <= 1e-3 against
        oracle/render.py, mask_rendered = depth > 0.2 bit-exact, flow labels and weights = lib/flow_c on the new depth (flags flip
        on < 1e-4 of the pixels: K·T may differ in the last ulp; flow <= 1e-4 elsewhere);
      * iteration 2 — forward + backward on the UPDATED batch with the UPDATED weights — against the oracle's training
        iteration on exactly that batch and those weights: net input bit-exact, losses <= 1e-4, all 42 gradients <= 2e-4."""
    from oracle import flow as oflow
    from oracle import render as orender
    from oracle import se3 as ose3
    from mx_deepim_amd.lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from mx_deepim_amd.lib.render_glumpy.render_py_multi import Render_Py
    B, H, W = 2, 480, 640
    d, cfg, net, params, data_np, label_np = _train_setup(ctx, B, 77, True)
    assert cfg.network.TRAIN_ITER_SIZE == 4        # yaml :58
    mesh = synthetic.ellipsoid_mesh([0.05, 0.04, 0.035], 24, 48)
    mesh.pop("uv")
    K = d["K"]
    rm = Render_Py("unused", ["obj"], K, W, H, meshes={"obj": mesh}, ctx=ctx, pixel_means=MEANS_REV.copy())
    upd = batchUpdaterPyMulti(cfg, H, W, render_machine=rm)
    data = {k: ctx.array(v) for k, v in data_np.items()}
    data.update(tgt_pose=ctx.array(d["pose_tgt"]), depth_gt_observed=ctx.array(d["depth_gt_observed"]))
    label = {k: ctx.array(v) for k, v in label_np.items()}
    snaps = []

    def on_iter(it, dat, lab):       # after update(): the batch this iteration ran on, its predictions, the weights it leaves
        snaps.append({"data": {k: v.asnumpy() for k, v in dat.items()}, "label": {k: v.asnumpy() for k, v in lab.items()},
                      "rot_est": net.act["rot_norm"].asnumpy(), "trans_est": net.act["trans_est"].asnumpy(),
                      "net_input": net.act["net_input"].asnumpy(), "pm_loss": float(net.act["pm_loss_sum"].asnumpy()[0]),
                      "flow_loss": float(net.act["flow_loss_sum"].asnumpy()[0]),
                      "grads": {k: v.asnumpy() for k, v in net.grad.items()} if it == 1 else None,
                      "params_before": params_before[0]})
        params_before[0] = {k: v.asnumpy() for k, v in net.params.items()}

    params_before = [{k: np.array(v) for k, v in params.items()}]
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)     # forward convs as single canonical chains: LeakyReLU kinks line up
    try:
        net.train_step(data, label, upd, iters=2, lr=1e-3, on_iter=on_iter)
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    s0, s1 = snaps
    # ---- the update between the iterations, from the GPU's own predictions
    mu, sd = cfg.dataset.trans_means, cfg.dataset.trans_stds
    src0 = s0["data"]["src_pose"]
    ref_pose = np.stack([ose3.RT_transform(src0[b], s0["rot_est"][b], s0["trans_est"][b], mu, sd, "CAMERA") for b in range(B)])
    pose1 = s1["data"]["src_pose"]
    np.testing.assert_allclose(pose1, ref_pose, rtol=1e-6, atol=1e-7)
    assert np.abs(pose1 - src0).max() > 1e-5                      # the pose really moved
    dep1 = s1["data"].get("depth_rendered")
    if dep1 is None:                                              # INPUT_DEPTH off: the depth is not a data entry; read the workspace
        dep1 = net._upd_ws[1]["depth_rendered"].asnumpy()
    for b in range(B):
        ri, rd = orender.render(mesh["vertices"], mesh["colors"], mesh["faces"], pose1[b], K, H, W, pixel_means=MEANS_REV)
        np.testing.assert_array_equal(dep1[b, 0], rd)
        np.testing.assert_allclose(s1["data"]["image_rendered"][b], ri, atol=1e-3)
    np.testing.assert_array_equal(s1["data"]["mask_rendered"], (dep1 > 0.2).astype(np.float32))
    # K·T (batch_updater_py_multi.py:255-259) against the oracle; then lib/flow_c on the GPU's own K·T bit for bit (F1 is
    # bit-exact given the same matrices — a last-ulp difference in K·T moves a flow value by ~1e-4 px and can flip a validity tie)
    KT_gpu = net._upd_ws[1]["KT"].asnumpy()
    np.testing.assert_allclose(KT_gpu, oflow.calc_KT(pose1, d["pose_tgt"], K), rtol=1e-5, atol=1e-5)
    rf, rv = oflow.gpu_flow(dep1, d["depth_gt_observed"], KT_gpu, np.linalg.inv(K).astype(np.float32))
    gv = s1["label"]["flow_weights"]
    np.testing.assert_array_equal(gv[:, :1], rv)
    np.testing.assert_array_equal(gv[:, 0], gv[:, 1])
    np.testing.assert_array_equal(s1["label"]["flow"], rf)
    assert np.count_nonzero(gv) > 1000
    # what the updater must NOT touch (training keeps them: data_pair.py's rectangle update is the TEST loop's)
    for k in ("image_observed", "mask_observed"):
        np.testing.assert_array_equal(s1["data"][k], s0["data"][k])
    for k in ("mask_gt_observed", "point_cloud_model", "point_cloud_observed", "point_cloud_weights"):
        np.testing.assert_array_equal(s1["label"][k], s0["label"][k])
    # the weights moved between the iterations
    assert not np.array_equal(s1["params_before"]["conv3_weight"], s0["params_before"]["conv3_weight"])
    # ---- iteration 2 against the oracle on the same batch and weights
    t = cfg.train_iter
    odata = {k: s1["data"][k] for k in ("image_observed", "image_rendered", "mask_observed", "mask_rendered", "src_pose")}
    ref_loss, g_ref, fwd = opipe.train_iteration(s1["params_before"], odata, s1["label"], K, MEANS_REV, mu, sd, cfg.network.ROT_COORD,
                                                 t.LW_PM, t.NUM_3D_SAMPLE, cfg.dataset.NORMALIZE_3D_POINT, t.SE3_PM_LOSS_TYPE,
                                                 t.SE3_PM_SL1_SCALAR, pred_flow=True, pred_mask=True, lw_flow=t.LW_FLOW,
                                                 lw_mask=t.LW_MASK, normalize_flow=cfg.dataset.NORMALIZE_FLOW)
    np.testing.assert_array_equal(s1["net_input"], fwd["net_input"])
    assert abs(s1["pm_loss"] - ref_loss) <= 1e-4 * abs(ref_loss)
    assert abs(s1["flow_loss"] - fwd["flow_loss_sum"]) <= 1e-4 * abs(fwd["flow_loss_sum"])
    assert set(s1["grads"]) == set(g_ref)
    for name in sorted(g_ref):
        if name.endswith("upsampling_weight"):
            assert not s1["grads"][name].any()
            continue
        close(s1["grads"][name], g_ref[name], 2e-4)
    # and the full step (TRAIN_ITER_SIZE = 4 iterations) runs resident: no allocation inside after the first call, finite losses
    seen = []
    net.train_step(data, label, upd, lr=1e-4, on_iter=lambda it, dat, lab: seen.append(float(net.act["pm_loss_sum"].asnumpy()[0])))
    assert len(seen) == 4 and all(np.isfinite(seen))


@pytest.mark.parametrize("pm_too,loss_type", [(True, "L2"), (False, "smooth_L1"), (True, "L1")])
def test_training_iteration_with_rot_and_trans_distance_losses(ctx, pm_too, loss_type):
    """train_iter.SE3_DIST_LOSS (deepIM_flownet.py:238-262): rot_loss = 1 - (q_gt . q_est)^2 (LW_ROT) on the normalised quaternion
    and the L2 / smooth_L1 / L1 translation loss (LW_TRANS) on the ZOOMED deltas, next to or instead of the point-matching loss;
    forward values and every parameter gradient against the oracle's backward."""
    B = 1
    d, cfg, net, params, data_np, label_np = _train_setup(ctx, B, 911, False)
    t = cfg.train_iter
    t.SE3_DIST_LOSS, t.SE3_PM_LOSS, t.LW_ROT, t.LW_TRANS, t.TRANS_LOSS_TYPE = True, pm_too, 0.7, 0.3, loss_type
    net = deepIM_flownet().get_symbol(cfg, is_train=True)
    net.bind_train(ctx, B, params, num_points=3000)
    from oracle import se3 as ose3
    rt = [ose3.calc_RT_delta(d["src_pose"][0][b], d["pose_tgt"][b], cfg.dataset.trans_means, cfg.dataset.trans_stds,
                             cfg.network.ROT_COORD, "QUAT") for b in range(B)]
    label_np["rot"] = np.stack([r for r, _ in rt]).astype(np.float32)
    label_np["trans"] = np.stack([tt for _, tt in rt]).astype(np.float32)
    data = {k: ctx.array(v) for k, v in data_np.items()}
    label = {k: ctx.array(v) for k, v in label_np.items()}
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)      # both sides differentiate at the same activations
    try:
        net.forward_train(data, label)
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    grads = net.backward()
    _, g_ref, fwd = opipe.train_iteration(params, data_np, label_np, d["K"], MEANS_REV, cfg.dataset.trans_means, cfg.dataset.trans_stds,
                                          cfg.network.ROT_COORD, t.LW_PM, t.NUM_3D_SAMPLE, cfg.dataset.NORMALIZE_3D_POINT,
                                          t.SE3_PM_LOSS_TYPE, t.SE3_PM_SL1_SCALAR, se3_pm_loss=pm_too, se3_dist_loss=True, lw_rot=t.LW_ROT,
                                          lw_trans=t.LW_TRANS, trans_loss_type=loss_type, trans_sigma=t.TRANS_SMOOTH_L1_SCALAR)
    np.testing.assert_array_equal(net.act["conv6_1"].asnumpy(), fwd["conv6_1"])
    close(net.act["zoom_trans_gt"].asnumpy(), fwd["zoom_trans_gt"], 1e-6)
    close(net.act["rot_loss"].asnumpy(), fwd["rot_loss"], 1e-5)
    close(net.act["trans_loss"].asnumpy().reshape(B, 3), fwd["trans_loss"].reshape(B, 3), 1e-5)
    assert float(fwd["rot_loss"].max()) > 1e-6 and float(np.abs(fwd["trans_loss"]).max()) > 1e-8
    for name in sorted(g_ref):
        assert np.abs(g_ref[name]).max() > 0, name
        close(grads[name].asnumpy(), g_ref[name], 2e-4)


def test_training_graph_refuses_what_the_reference_graph_does_not_build(ctx):
    cfg = default_config()
    cfg.network.PRED_FLOW = cfg.network.PRED_MASK = False
    cfg.network.ROT_TYPE = "EULER"
    with pytest.raises(NotImplementedError):
        deepIM_flownet().get_symbol(cfg, is_train=True)
    cfg.network.ROT_TYPE = "QUAT"
    cfg.train_iter.SE3_DIST_LOSS, cfg.train_iter.TRANS_LOSS_TYPE = True, "huber"
    with pytest.raises(Exception, match="TRANS_LOSS_TYPE"):
        deepIM_flownet().get_symbol(cfg, is_train=True)
    cfg.train_iter.SE3_DIST_LOSS, cfg.train_iter.SE3_PM_LOSS = False, False
    with pytest.raises(NotImplementedError):
        deepIM_flownet().get_symbol(cfg, is_train=True)
