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


def test_training_iteration_of_the_pose_branch_matches_oracle(ctx):
    """One training-style iteration (B = 1, 480x640): zoom from the gt mask → encoder → fc → rot/trans → Transform3D →
    point-matching loss, backward through everything, SGD step (module.py:1131-1137 order)."""
    B = 1
    d = synthetic.make_batch(B, seed=910, n_frames=1)
    cfg = default_config()
    cfg.network.PRED_FLOW = cfg.network.PRED_MASK = False
    net = deepIM_flownet().get_symbol(cfg, is_train=True)
    params = net.init_weights(cfg, seed=91)
    net.bind_train(ctx, B, params, num_points=3000)
    gt = (d["depth_gt_observed"] > 0).astype(np.float32)
    pco = np.stack([d["pose_tgt"][b][:, :3].astype(np.float64) @ d["point_cloud_model"][b].astype(np.float64) + d["pose_tgt"][b][:, 3:4]
                    for b in range(B)]).astype(np.float32)
    wts = np.ones((B, 3, 3000), np.float32)
    data_np = {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][0], "mask_observed": d["mask_observed"],
               "mask_rendered": d["mask_rendered"][0], "src_pose": d["src_pose"][0]}
    label_np = {"mask_gt_observed": gt, "point_cloud_model": d["point_cloud_model"], "point_cloud_weights": wts,
                "point_cloud_observed": pco}
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


def test_train_symbol_refuses_the_heads_it_cannot_backpropagate():
    cfg = default_config()
    with pytest.raises(NotImplementedError):
        deepIM_flownet().get_symbol(cfg, is_train=True)
