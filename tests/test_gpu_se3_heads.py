"""S- and H-group parity on the GPU (C ABI) against the CPU oracle."""
import ctypes

import numpy as np
import pytest

from oracle import heads as oh
from oracle import se3 as ose3
from mx_deepim_amd.runtime import lib
from mx_deepim_amd import synthetic

pytestmark = pytest.mark.gpu
cf = ctypes.c_float
COORDS = {"MODEL": 0, "CAMERA": 1, "CAMERA_NEW": 2, "NAIVE": 3}


def _poses(rng, B):
    return np.stack([synthetic.sample_pose_pair(rng)[1] for _ in range(B)]).astype(np.float32)


@pytest.mark.parametrize("coord", list(COORDS))
def test_rt_transform(ctx, coord):
    rng = np.random.default_rng(1)
    B = 37
    src = _poses(rng, B)
    se3 = np.concatenate([rng.standard_normal((B, 4)) * 0.3 + [1, 0, 0, 0], rng.standard_normal((B, 3)) * 0.1], 1).astype(np.float32)
    mu, sd = np.array([0.01, -0.02, 0.03], np.float32), np.array([0.9, 1.1, 1.2], np.float32)
    ref = np.stack([ose3.RT_transform(src[b], se3[b, :4], se3[b, 4:], mu, sd, coord) for b in range(B)])
    out, out64 = ctx.empty((B, 3, 4)), ctx.empty((B, 3, 4), dtype=np.float64)
    lib.deepim_rt_transform(ctx.handle, out, out64, ctx.array(src), ctx.array(se3), mu, sd, COORDS[coord], B)
    np.testing.assert_allclose(out64.asnumpy(), ref, rtol=1e-6, atol=1e-7)   # north_star bar is 1e-4
    np.testing.assert_allclose(out.asnumpy(), ref.astype(np.float32), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("coord", list(COORDS))
def test_transform3d_forward_backward(ctx, coord):
    rng = np.random.default_rng(2)
    B, N = 8, 3000
    pts = rng.standard_normal((B, 3, N)).astype(np.float32) * 0.05
    q = rng.standard_normal((B, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0] = [2, 0, 0, 0]  # un-normalised → identity branch (transform3d.py:188-189), zero quaternion grad
    t = (rng.standard_normal((B, 3)) * 0.05).astype(np.float32)
    src = _poses(rng, B)
    mu, sd = np.zeros(3, np.float32), np.ones(3, np.float32)
    ref = ose3.transform3d_forward(pts, q, t, src, mu, sd, coord)
    out = ctx.empty((B, 3, N))
    dp, dq_, dt_, dsrc = ctx.array(pts), ctx.array(q), ctx.array(t), ctx.array(src)
    lib.deepim_transform3d_forward(ctx.handle, out, dp, dq_, dt_, dsrc, mu, sd, COORDS[coord], B, N)
    np.testing.assert_allclose(out.asnumpy(), ref, rtol=1e-6, atol=1e-7)
    og = rng.standard_normal((B, 3, N)).astype(np.float32)
    rq, rt = ose3.transform3d_backward(og, pts, q, t, src, mu, sd, coord)
    gq, gt = ctx.empty((B, 4)), ctx.empty((B, 3))
    lib.deepim_transform3d_backward(ctx.handle, gq, gt, ctx.array(og), dp, dq_, dt_, dsrc, mu, sd, COORDS[coord], B, N)
    scale_q, scale_t = np.abs(rq).max() + 1e-6, np.abs(rt).max() + 1e-6
    assert np.abs(gq.asnumpy() - rq).max() / scale_q < 1e-4
    assert np.abs(gt.asnumpy() - rt).max() / scale_t < 1e-4
    assert not gq.asnumpy()[0].any()


@pytest.mark.parametrize("lt", [("L1", 0), ("L2", 1), ("smooth_L1", 2)])
def test_point_matching_loss(ctx, lt):
    rng = np.random.default_rng(3)
    B, N = 4, 3000
    est = rng.standard_normal((B, 3, N)).astype(np.float32) * 0.1
    gt = rng.standard_normal((B, 3, N)).astype(np.float32) * 0.1
    w = (rng.random((B, 3, N)) > 0.2).astype(np.float32)
    rl, rs, rg = oh.point_matching_loss(est, gt, w, 0.1, lt[0], 1.5, 0.1 / 3000)
    loss, s, g = ctx.empty(est.shape), ctx.empty((1,)), ctx.empty(est.shape)
    lib.deepim_point_matching_loss(ctx.handle, loss, s, g, ctx.array(est), ctx.array(gt), ctx.array(w), cf(0.1), lt[1],
                                   cf(1.5), cf(0.1 / 3000), B, N)
    np.testing.assert_allclose(loss.asnumpy(), rl, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(g.asnumpy(), rg, rtol=1e-6, atol=1e-9)
    assert abs(float(s.asnumpy()[0]) - rs) / rs < 1e-5


def test_flow_loss_and_logistic(ctx):
    rng = np.random.default_rng(4)
    n = 2 * 2 * 97 * 131
    est, gt = rng.standard_normal(n).astype(np.float32), (rng.standard_normal(n) * 20).astype(np.float32)
    w = (rng.random(n) > 0.5).astype(np.float32)
    rl, rs, rg = oh.flow_loss(est, gt, w, 20.0, 0.25 / (480 * 640))
    loss, s, g = ctx.empty((n,)), ctx.empty((1,)), ctx.empty((n,))
    lib.deepim_flow_loss(ctx.handle, loss, s, g, ctx.array(est), ctx.array(gt), ctx.array(w), cf(20.0),
                         cf(0.25 / (480 * 640)), n)
    np.testing.assert_allclose(loss.asnumpy(), rl, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(g.asnumpy(), rg, rtol=1e-6, atol=1e-12)
    assert abs(float(s.asnumpy()[0]) - rs) / rs < 1e-5
    lab = (rng.random(n) > 0.5).astype(np.float32)
    rp, rgl = oh.mask_logistic(est * 3, lab, 0.03)
    p, gl = ctx.empty((n,)), ctx.empty((n,))
    lib.deepim_mask_logistic(ctx.handle, p, gl, ctx.array(est * 3), ctx.array(lab), cf(0.03), n)
    np.testing.assert_allclose(p.asnumpy(), rp, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(gl.asnumpy(), rgl, rtol=1e-5, atol=1e-8)


def test_group_picker(ctx):
    rng = np.random.default_rng(5)
    B, C, G = 6, 52, 13
    x = rng.standard_normal((B, C)).astype(np.float32)
    idx = rng.integers(0, G, (B,)).astype(np.float32)
    out = ctx.empty((B, C // G))
    lib.deepim_group_picker_forward(ctx.handle, out, ctx.array(x), ctx.array(idx), G, B, C)
    np.testing.assert_array_equal(out.asnumpy(), oh.group_picker(x, idx, G))
    og = rng.standard_normal((B, C // G)).astype(np.float32)
    gin = ctx.empty((B, C))
    lib.deepim_group_picker_backward(ctx.handle, gin, ctx.array(og), ctx.array(idx), G, B, C)
    np.testing.assert_array_equal(gin.asnumpy(), oh.group_picker_backward(og, idx, G, C))


@pytest.mark.parametrize("coord", list(COORDS))
def test_calc_rt_delta(ctx, coord):
    """Ground-truth labels (calc_RT_delta, rot_type QUAT) vs the oracle, which is pinned to the reference."""
    rng = np.random.default_rng(6)
    B = 33
    src, tgt = _poses(rng, B), _poses(rng, B)
    mu, sd = np.array([0.01, -0.02, 0.03], np.float32), np.array([0.9, 1.1, 1.2], np.float32)
    rot, trans = ctx.empty((B, 4)), ctx.empty((B, 3))
    lib.deepim_calc_rt_delta(ctx.handle, rot, trans, ctx.array(src), ctx.array(tgt), mu, sd, COORDS[coord], B)
    for b in range(B):
        q, t = ose3.calc_RT_delta(src[b], tgt[b], mu.astype(np.float64), sd.astype(np.float64), coord, "QUAT")
        np.testing.assert_allclose(rot.asnumpy()[b], q, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(trans.asnumpy()[b], t, rtol=1e-5, atol=1e-6)
    # round trip: applying the delta to the source pose reproduces the target (north_star 1e-4 bar)
    se3 = np.concatenate([rot.asnumpy(), trans.asnumpy()], 1).astype(np.float32)
    out = ctx.empty((B, 3, 4))
    lib.deepim_rt_transform(ctx.handle, out, None, ctx.array(src), ctx.array(se3), mu, sd, COORDS[coord], B)
    np.testing.assert_allclose(out.asnumpy(), tgt, rtol=1e-4, atol=1e-5)


def test_l2_normalize_and_rot_dist_loss(ctx):
    rng = np.random.default_rng(7)
    B = 19
    x = rng.standard_normal((B, 4)).astype(np.float32)
    g = rng.standard_normal((B, 4)).astype(np.float32)
    out, gin = ctx.empty((B, 4)), ctx.empty((B, 4))
    lib.deepim_l2_normalize_forward(ctx.handle, out, ctx.array(x), B, 4, cf(1e-10))
    np.testing.assert_allclose(out.asnumpy(), oh.l2_normalize(x), rtol=1e-6, atol=1e-7)
    lib.deepim_l2_normalize_backward(ctx.handle, gin, ctx.array(g), ctx.array(x), B, 4, cf(1e-10))
    np.testing.assert_allclose(gin.asnumpy(), oh.l2_normalize_backward(g, x), rtol=1e-5, atol=1e-6)
    qg = oh.l2_normalize(rng.standard_normal((B, 4)).astype(np.float32))
    qe = oh.l2_normalize(x)
    loss, dq = ctx.empty((B,)), ctx.empty((B, 4))
    lib.deepim_rot_dist_loss(ctx.handle, loss, dq, ctx.array(qg), ctx.array(qe), cf(0.5), B)
    rl, rd = oh.rot_dist_loss(qg, qe, 0.5)
    np.testing.assert_allclose(loss.asnumpy(), rl, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dq.asnumpy(), rd, rtol=1e-5, atol=1e-6)
    # translation distance loss = point_matching_loss with N=1, normalize 1 (deepIM_flownet.py:250-262)
    est, gt = rng.standard_normal((B, 3, 1)).astype(np.float32), rng.standard_normal((B, 3, 1)).astype(np.float32)
    l, s, d = ctx.empty(est.shape), ctx.empty((1,)), ctx.empty(est.shape)
    lib.deepim_point_matching_loss(ctx.handle, l, s, d, ctx.array(est), ctx.array(gt), None, cf(1.0), 2, cf(3.0), cf(1.0), B, 1)
    rl2, _, rg2 = oh.point_matching_loss(est, gt, None, 1.0, "smooth_L1", 3.0, 1.0)
    np.testing.assert_allclose(l.asnumpy(), rl2, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(d.asnumpy(), rg2, rtol=1e-6, atol=1e-7)


def test_pose_error_metrics_match_reference(ctx):
    """re / te / ADD / ADI / 2-D reprojection error vs the reference's own lib/utils/pose_error.py
    (golden vectors generated from /root/reference, tests/golden/make_golden.py:pose_error_golden)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_error_golden.npz"))
    B, _, N = g["points"].shape
    out = ctx.empty((B, 5))
    lib.deepim_pose_error(ctx.handle, out, ctx.array(g["pose_est"]), ctx.array(g["pose_gt"]), ctx.array(g["points"]), 0,
                          np.ascontiguousarray(g["K"]), B, N)
    np.testing.assert_allclose(out.asnumpy(), g["metrics"], rtol=2e-4, atol=1e-6)
    # shared point set + identical poses → all metrics zero
    lib.deepim_pose_error(ctx.handle, out, ctx.array(g["pose_gt"]), ctx.array(g["pose_gt"]), ctx.array(g["points"][0]), 1,
                          np.ascontiguousarray(g["K"]), B, N)
    z = out.asnumpy()
    assert np.abs(z[:, 1:]).max() < 1e-6 and z[:, 0].max() < 0.05   # acos near 1: float32 rotation matrices are not exactly orthonormal
