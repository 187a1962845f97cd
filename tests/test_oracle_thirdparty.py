"""Cross-checks of the oracle's restatement of third-party MXNet operator semantics (unpinned by the
reference's own tests) against torch-CPU, used here only as an independent second opinion."""
import numpy as np
import pytest

from oracle import net as onet
from oracle import zoom as oz

torch = pytest.importorskip("torch")
F = torch.nn.functional


def test_sampler_matches_torch_grid_sample():
    rng = np.random.default_rng(0)
    H, W = 48, 64
    img = rng.standard_normal((3, H, W)).astype(np.float32)
    for wx, wy, tx, ty in [(1, 1, 0, 0), (0.3, 0.3, 0.1, -0.2), (0.17, 0.17, 0.9, 0.8), (2.5, 2.5, -0.3, 0.4)]:
        got = oz.bilinear_sample(img, np.float32(wx), np.float32(wy), np.float32(tx), np.float32(ty))
        theta = torch.tensor([[[wx, 0, tx], [0, wy, ty]]], dtype=torch.float32)
        grid = F.affine_grid(theta, (1, 3, H, W), align_corners=True)
        ref = F.grid_sample(torch.from_numpy(img)[None], grid, mode="bilinear", padding_mode="zeros", align_corners=True)
        np.testing.assert_allclose(got, ref[0].numpy(), rtol=1e-4, atol=2e-4)


def test_identity_zoom_indices():
    idx = oz.sample_indices(np.array([[1, 1, 0, 0]], np.float32), 480, 640)
    x = np.arange(640)
    # the f32 grid may land an ulp below the pixel centre: floor index is x or x-1 with weight ≈ 0
    assert np.all((idx[0, 0, 0] == x) | (idx[0, 0, 0] == x - 1))


def test_roundf_is_half_away_from_zero():
    np.testing.assert_array_equal(oz.roundf(np.array([0.5, 1.5, 2.5, -0.5, -1.5, 0.49999997], np.float32)),
                                  np.array([1, 2, 3, -1, -2, 0], np.float32))


def test_conv_deconv_upsample_fc_match_torch():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((2, 6, 21, 29)).astype(np.float32)
    for k, s, p in ((7, 2, 3), (5, 2, 2), (3, 1, 1), (3, 2, 1)):
        w = rng.standard_normal((9, 6, k, k)).astype(np.float32) / (k * 3)
        b = rng.standard_normal(9).astype(np.float32)
        ref = F.leaky_relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=s, padding=p), 0.1)
        np.testing.assert_allclose(onet.conv2d(x, w, b, s, p, 0.1), ref.numpy(), rtol=1e-4, atol=1e-5)
    wd = rng.standard_normal((6, 4, 4, 4)).astype(np.float32) / 5
    bd = rng.standard_normal(4).astype(np.float32)
    full = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(wd), torch.from_numpy(bd), stride=2).numpy()
    np.testing.assert_allclose(onet.deconv4x4s2_crop(x, wd, bd, 40, 56, (1, 1), 1.0), full[:, :, 1:41, 1:57], rtol=1e-4,
                               atol=1e-5)
    wu = onet.bilinear_upsample_weights(6)
    xs = x[:, :, :5, :6]
    full = F.conv_transpose2d(torch.from_numpy(xs), torch.from_numpy(wu), stride=16, groups=6).numpy()
    np.testing.assert_allclose(onet.upsample16_crop(xs, wu, 80, 96, (8, 8), 20.0), 20 * full[:, :, 8:88, 8:104],
                               rtol=1e-4, atol=1e-4)
    xf = rng.standard_normal((3, 200)).astype(np.float32)
    wf = rng.standard_normal((7, 200)).astype(np.float32)
    bf = rng.standard_normal(7).astype(np.float32)
    np.testing.assert_allclose(onet.fc(xf, wf, bf, 0.1),
                               F.leaky_relu(F.linear(torch.from_numpy(xf), torch.from_numpy(wf), torch.from_numpy(bf)), 0.1).numpy(),
                               rtol=1e-4, atol=1e-5)


def test_bilinear_init_is_separable_tent():
    w = onet.bilinear_upsample_weights(2)
    assert w.shape == (2, 1, 32, 32)
    k1 = 1 - np.abs(np.arange(32) / 16.0 - (2 * 16 - 1) / 32.0)
    np.testing.assert_allclose(w[1, 0], np.outer(k1, k1), rtol=1e-6)


def test_conv_order_switch_matches_torch_and_itself():
    """Both accumulation orders of the conv restatement are the same convolution (torch as second opinion); they differ
    from each other only by fp32 re-association, and coincide exactly when every product is exact."""
    rng = np.random.default_rng(21)
    x = rng.standard_normal((2, 6, 13, 17)).astype(np.float32)
    w = (rng.standard_normal((5, 6, 3, 3)) / 7).astype(np.float32)
    b = rng.standard_normal(5).astype(np.float32)
    ref = F.leaky_relu(F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=2, padding=1), 0.1).numpy()
    a = onet.conv2d(x, w, b, 2, 1, 0.1)
    p = onet.conv2d(x, w, b, 2, 1, 0.1, pair_order=True)
    np.testing.assert_allclose(a, ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(p, ref, rtol=1e-4, atol=1e-5)
    assert np.abs(a - p).max() < 1e-5
    xi = rng.integers(-3, 4, x.shape).astype(np.float32)     # small integers: every partial sum is exact in fp32
    wi = rng.integers(-3, 4, w.shape).astype(np.float32)
    np.testing.assert_array_equal(onet.conv2d(xi, wi, None, 1, 1, 1.0), onet.conv2d(xi, wi, None, 1, 1, 1.0, pair_order=True))


@pytest.mark.parametrize("case", [(2, 8, 33, 41, 64, 7, 2, 3), (1, 13, 16, 20, 21, 5, 2, 2), (2, 16, 12, 14, 24, 3, 1, 1),
                                  (1, 64, 30, 40, 128, 5, 2, 2), (1, 5, 9, 11, 3, 3, 1, 1)])
def test_blocked_conv_build_is_bit_identical_to_the_checker(case):
    """oracle_conv2d_blocked — the cache-blocked OpenMP build bench.py's `cpu_baseline` times — runs, per output, the same
    float32 fmaf chain over (ci,ky,kx) as the checker oracle_conv2d: equal bit for bit, ragged channel blocks and borders
    included."""
    B, Cin, H, W, Cout, k, s, p = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, k, k)) * 0.1).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    want = onet.conv2d(x, w, b, s, p, 0.1)
    onet.BLOCKED = True
    try:
        got = onet.conv2d(x, w, b, s, p, 0.1)
    finally:
        onet.BLOCKED = False
    np.testing.assert_array_equal(got, want)
    assert onet.omp_threads() >= 1


def test_oracle_backward_matches_torch_autograd():
    """oracle/net.c conv / FC backward and the LeakyReLU / SGD helpers against torch autograd / torch.optim.SGD-style math."""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    from oracle import net as onet
    rng = np.random.default_rng(0)
    for (B, cin, H, W, cout, k, s, p) in [(2, 3, 9, 11, 4, 3, 1, 1), (1, 4, 12, 10, 5, 5, 2, 2), (2, 2, 13, 9, 3, 7, 2, 3), (1, 6, 8, 10, 4, 3, 2, 1)]:
        x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
        w = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        xt, wt, bt = (torch.tensor(a, requires_grad=True) for a in (x, w, b))
        y = F.leaky_relu(F.conv2d(xt, wt, bt, stride=s, padding=p), 0.1)
        dy = rng.standard_normal(tuple(y.shape)).astype(np.float32)
        y.backward(torch.tensor(dy))
        dz = onet.lrelu_backward(dy, y.detach().numpy(), 0.1)
        dx, dw, db = onet.conv2d_backward(x, w, dz, s, p)
        for got, ref in ((dx, xt.grad), (dw, wt.grad), (db, bt.grad)):
            np.testing.assert_allclose(got, ref.numpy(), rtol=1e-4, atol=2e-5)
    x, w, dy = (rng.standard_normal(sh).astype(np.float32) for sh in ((3, 20), (5, 20), (3, 5)))
    xt, wt, bt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True), torch.zeros(5, requires_grad=True)
    F.linear(xt, wt, bt).backward(torch.tensor(dy))
    for got, ref in zip(onet.fc_backward(x, w, dy), (xt.grad, wt.grad, bt.grad)):
        np.testing.assert_allclose(got, ref.numpy(), rtol=1e-5, atol=1e-6)
    # MXNet sgd_mom_update == torch SGD with the learning rate folded into the momentum buffer
    wv, g = rng.standard_normal(50).astype(np.float32), rng.standard_normal(50).astype(np.float32)
    w1, m1 = onet.sgd_mom_update(wv, np.zeros(50, np.float32), g, 0.1, 0.01, 0.9)
    w2, _ = onet.sgd_mom_update(w1, m1, g, 0.1, 0.01, 0.9)
    pt = torch.tensor(wv, requires_grad=True)
    opt = torch.optim.SGD([pt], lr=0.1, momentum=0.9, weight_decay=0.01)
    for _ in range(2):
        pt.grad = torch.tensor(g)
        opt.step()
    np.testing.assert_allclose(w2, pt.detach().numpy(), rtol=1e-5, atol=1e-6)


def test_oracle_decoder_backward_matches_torch_autograd():
    """oracle/net.c Deconvolution(k4 s2)+Crop and the depthwise k32 s16 upsampling + Crop backward against torch autograd."""
    rng = np.random.default_rng(12)
    for B, cin, H, W, cout, ho, wo in [(2, 5, 6, 7, 3, 13, 15), (1, 2, 8, 10, 2, 15, 20)]:
        x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
        w = rng.standard_normal((cin, cout, 4, 4)).astype(np.float32)
        dy = rng.standard_normal((B, cout, ho, wo)).astype(np.float32)
        xt, wt, bt = torch.tensor(x, requires_grad=True), torch.tensor(w, requires_grad=True), torch.zeros(cout, requires_grad=True)
        y = F.conv_transpose2d(xt, wt, bt, stride=2)[:, :, 1:1 + ho, 1:1 + wo]
        y.backward(torch.tensor(dy))
        dx, dw, db = onet.deconv4x4s2_crop_backward(x, w, dy, (1, 1))
        np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(dw, wt.grad.numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-4, atol=1e-4)
    B, C, H, W, Ho, Wo = 2, 2, 5, 6, 70, 90
    wu = rng.standard_normal((C, 1, 32, 32)).astype(np.float32)
    dy = rng.standard_normal((B, C, Ho, Wo)).astype(np.float32)
    xt = torch.zeros((B, C, H, W), requires_grad=True)
    y = 1.5 * F.conv_transpose2d(xt, torch.tensor(wu), stride=16, groups=C)[:, :, 8:8 + Ho, 8:8 + Wo]
    y.backward(torch.tensor(dy))
    np.testing.assert_allclose(onet.upsample16_crop_backward(dy, wu, H, W, (8, 8), 1.5), xt.grad.numpy(), rtol=1e-4, atol=1e-4)
