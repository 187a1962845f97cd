"""fp32 Winograd F(4,3) x F(2,3) layers (csrc/wino42.hip: 4-row x 2-column output tiles, 24 positions) against the C oracle's direct
convolution: a different summation of other fp32 products, so the bar is a tolerance — 1e-5 of the layer's output range, as for the
F(2x2,3x3) kernels (tests/test_gpu_wino.py)."""
import ctypes

import numpy as np
import pytest

from oracle import net as onet
from mx_deepim_amd.runtime import DeviceArray, lib

pytestmark = pytest.mark.gpu
cf = ctypes.c_float
TOL = 1e-5


def _to_nc8(x):
    B, C, H, W = x.shape
    return np.ascontiguousarray(x.reshape(B, C // 8, 8, H, W).transpose(0, 1, 3, 4, 2))


def _from_nc8(y, shape):
    B, C, H, W = shape
    return np.ascontiguousarray(y.reshape(B, C // 8, H, W, 8).transpose(0, 1, 4, 2, 3).reshape(B, C, H, W))


def _pack(ctx, w):
    cout, cin = w.shape[:2]
    pk = DeviceArray(ctx, (lib.load().deepim_conv_wino42_packed_size(cout, cin) // 4,))
    lib.deepim_conv_wino42_pack_weights(ctx.handle, pk, ctx.array(w), cout, cin)
    return pk


# (B, Cin, H, W, Cout): encoder geometries at reduced size, H % 4 != 0 and odd W (half-covered tiles), single tiles, ragged tile blocks,
# one / several channel blocks (XCD deal: gy = 1, 2, 3, 4, 8, 16)
CASES = [
    (2, 256, 12, 16, 256),     # conv3_1 channels
    (1, 512, 30, 40, 512),     # conv4_1 at full spatial size, one sample (30 rows = 7.5 tiles)
    (3, 64, 15, 20, 64),       # odd H
    (2, 1024, 8, 10, 1024),    # conv6_1 geometry
    (5, 8, 7, 9, 64),          # odd H and W, one step
    (1, 16, 2, 2, 192),        # a single tile, 3 channel blocks
    (2, 24, 1, 5, 128),        # one row: every tile is three quarters outside
    (33, 8, 6, 6, 64),         # 198 tiles: ragged last block of 32
    (1, 40, 60, 80, 128),      # conv3_1's frame, 5 steps (odd step count)
]


@pytest.mark.parametrize("case", CASES)
def test_wino42_layer_within_1e5_of_the_direct_convolution(ctx, case):
    B, cin, H, W, cout = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    x *= rng.uniform(size=x.shape) > 0.3                      # post-LeakyReLU-like sparsity is irrelevant to the bound; keep zeros in
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    want = onet.conv2d(x, w, b, 1, 1, 0.1)
    xin, pk, bias = ctx.array(_to_nc8(x)), _pack(ctx, w), ctx.array(b)
    scale = max(1.0, float(np.abs(want).max()))
    # channel-blocked output
    out = ctx.array(np.full((B, cout, H, W), 7.0, np.float32))
    lib.deepim_conv2d_wino42_forward(ctx.handle, out, xin, pk, bias, B, cin, H, W, cout, cf(0.1), 1, 0, 0)
    got = _from_nc8(out.asnumpy(), (B, cout, H, W))
    err = float(np.abs(got - want).max() / scale)
    print("F(4,3)xF(2,3) %s: %.2e of range" % (case, err))
    assert err <= TOL
    # NCHW into a channel slice of a wider tensor
    o2 = ctx.array(np.full((B, cout + 16, H, W), 3.0, np.float32))
    lib.deepim_conv2d_wino42_forward(ctx.handle, o2, xin, pk, bias, B, cin, H, W, cout, cf(0.1), 0, cout + 16, 8)
    g2 = o2.asnumpy()
    assert np.abs(g2[:, 8:8 + cout] - want).max() / scale <= TOL
    assert (g2[:, :8] == 3.0).all() and (g2[:, 8 + cout:] == 3.0).all()
    np.testing.assert_array_equal(g2[:, 8:8 + cout], got)      # the two output forms hold the same numbers


def test_wino42_weight_transform_is_G4_g_G2t_with_the_last_column_negated(ctx):
    rng = np.random.default_rng(5)
    cout, cin = 64, 16
    w = rng.standard_normal((cout, cin, 3, 3)).astype(np.float32)
    pk = _pack(ctx, w).asnumpy().reshape(cout // 32, cin // 8, 24, 2, 32, 4)      # [mb][c8][pos][h][row][s]
    from oracle import wino
    U = np.einsum("xa,ocab,nb->ocxn", wino.G4, w.astype(np.float64), wino.G)
    U[..., 3] *= -1
    for mb in range(cout // 32):
        for c8 in range(cin // 8):
            for h in range(2):
                for s in range(4):
                    ci = c8 * 8 + 4 * h + s
                    want = U[mb * 32:(mb + 1) * 32, ci].reshape(32, 24).T.astype(np.float32)     # [pos][row]
                    np.testing.assert_array_equal(pk[mb, c8, :, h, :, s], want)


def test_wino42_argument_checks(ctx):
    L = lib.load()
    assert L.deepim_conv_wino42_packed_size(64, 8) == 64 * 8 * 24 * 4
    assert L.deepim_conv_wino42_packed_size(32, 8) == 0 and L.deepim_conv_wino42_packed_size(64, 12) == 0
    x = ctx.zeros((1, 1, 4, 4, 8))
    with pytest.raises(RuntimeError):
        lib.deepim_conv2d_wino42_forward(ctx.handle, x, x, x, None, 1, 8, 4, 4, 32, cf(0.1), 1, 0, 0)      # Cout % 64
