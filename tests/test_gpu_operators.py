"""The CustomOp mirrors driven through the mx shim (`mx.nd.Custom(op_type=...)`, string attrs, req
handling) on the GPU — the drop-in boundary B1 of SURVEY §8b."""
import numpy as np
import pytest

import mx_deepim_amd.operator_py  # noqa: F401
from mx_deepim_amd import mx, synthetic
from mx_deepim_amd.lib.flow_c.flow import gpu_flow, gpu_flow_wrapper
from mx_deepim_amd.lib.pair_matching import RT_transform as dRT
from mx_deepim_amd.lib.pair_matching.flow import calc_flow as d_calc_flow
from oracle import flow as oflow
from oracle import se3 as ose3
from oracle import zoom as oz

pytestmark = pytest.mark.gpu
MEANS_CFG = synthetic.PIXEL_MEANS  # config order; the Props reverse it


def test_zoom_ops_through_custom(ctx, small_batch):
    d = small_batch
    K = d["K"].flatten()
    nd = lambda a: mx.nd.array(a, ctx=ctx)  # noqa: E731
    zmo, zmg, zmr, zf = mx.nd.Custom(mask_observed=nd(d["mask_observed"]), mask_gt_observed=nd(d["mask_observed"]),
                                     mask_rendered=nd(d["mask_rendered"][0]), src_pose=nd(d["src_pose"][0]), K=K,
                                     name="ZoomMask", op_type="ZoomMask", height=480, width=640)
    r0, _, r2, rzf = oz.zoom_mask(d["mask_observed"], d["mask_observed"], d["mask_rendered"][0], d["src_pose"][0], d["K"])
    np.testing.assert_array_equal(zf.asnumpy(), rzf)
    np.testing.assert_array_equal(zmo.asnumpy(), r0)
    np.testing.assert_array_equal(zmr.asnumpy(), r2)
    zio, zir = mx.nd.Custom(zoom_factor=zf, image_observed=nd(d["image_observed"]), image_rendered=nd(d["image_rendered"][0]),
                            op_type="ZoomImageWithFactor", height=480, width=640, pixel_means=MEANS_CFG.flatten())
    q0, q1 = oz.zoom_image_with_factor(rzf, d["image_observed"], d["image_rendered"][0], MEANS_CFG[::-1])
    np.testing.assert_array_equal(zio.asnumpy(), q0)
    np.testing.assert_array_equal(zir.asnumpy(), q1)
    t = np.random.default_rng(0).standard_normal((2, 3)).astype(np.float32)
    zt = mx.nd.Custom(zoom_factor=zf, trans_delta=nd(t), op_type="ZoomTrans", b_inv_zoom=True)
    np.testing.assert_array_equal(zt.asnumpy(), oz.zoom_trans(rzf, t, True))
    # empty observed mask → the reference's ValueError
    empty = nd(np.zeros_like(d["mask_observed"]))
    with pytest.raises(ValueError):
        mx.nd.Custom(empty, empty, empty, nd(d["src_pose"][0]), K=K, op_type="ZoomMask")


def test_assign_req_add_and_null(ctx):
    op = mx.operator.get_registered("ZoomTrans")(b_inv_zoom="True").create_operator(ctx, None, None)
    zf = mx.nd.array(np.array([[0.5, 0.5, 0, 0]], np.float32), ctx=ctx)
    t = mx.nd.array(np.array([[2, 4, 6]], np.float32), ctx=ctx)
    out = mx.nd.array(np.array([[10, 10, 10]], np.float32), ctx=ctx)
    op.forward(False, ["add"], [zf, t], [out], [])
    np.testing.assert_array_equal(out.asnumpy(), [[11, 12, 16]])
    op.forward(False, ["null"], [zf, t], [out], [])
    np.testing.assert_array_equal(out.asnumpy(), [[11, 12, 16]])


def test_transform3d_operator_forward_backward(ctx):
    rng = np.random.default_rng(1)
    B, N = 4, 3000
    pts = (rng.standard_normal((B, 3, N)) * 0.05).astype(np.float32)
    q = rng.standard_normal((B, 4))
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    t = (rng.standard_normal((B, 3)) * 0.05).astype(np.float32)
    src = np.stack([synthetic.sample_pose_pair(rng)[1] for _ in range(B)])
    nd = lambda a: mx.nd.array(a, ctx=ctx)  # noqa: E731
    ins = [nd(pts), nd(q), nd(t), nd(src)]
    out = mx.nd.Custom(*ins, op_type="Transform3D", T_means=np.zeros(3), T_stds=np.ones(3), rot_coord="CAMERA")
    np.testing.assert_allclose(out.asnumpy(), ose3.transform3d_forward(pts, q, t, src, np.zeros(3), np.ones(3), "CAMERA"),
                               rtol=1e-6, atol=1e-7)
    op = mx.nd.Custom.last_operator
    og = rng.standard_normal((B, 3, N)).astype(np.float32)
    grads = [mx.nd.zeros(a.shape, ctx=ctx) for a in ins]
    op.backward(["write"] * 4, [nd(og)], ins, [out], grads, [])
    rq, rt = ose3.transform3d_backward(og, pts, q, t, src, np.zeros(3), np.ones(3), "CAMERA")
    assert np.abs(grads[1].asnumpy() - rq).max() / np.abs(rq).max() < 1e-4
    assert np.abs(grads[2].asnumpy() - rt).max() / np.abs(rt).max() < 1e-4
    assert not grads[0].asnumpy().any() and not grads[3].asnumpy().any()


def test_lib_flow_c_and_pair_matching_entries(ctx, small_batch):
    d = small_batch
    KT = oflow.calc_KT(d["src_pose"][0], d["pose_tgt"], d["K"])
    Kinv = np.linalg.inv(d["K"]).astype(np.float32)
    flow, valid = gpu_flow_wrapper(0)(d["depth_rendered"][0], d["depth_gt_observed"], KT, Kinv)
    rf, rv = oflow.gpu_flow(d["depth_rendered"][0], d["depth_gt_observed"], KT, Kinv)
    np.testing.assert_array_equal(flow, rf)
    np.testing.assert_array_equal(valid, rv)
    f2, v2 = gpu_flow(d["depth_rendered"][0][:1], d["depth_gt_observed"][:1], KT[:1], Kinv)
    np.testing.assert_array_equal(f2, rf[:1])
    np.testing.assert_allclose(dRT.calc_KT(d["src_pose"][0], d["pose_tgt"], d["K"], ctx).asnumpy(), KT, rtol=1e-6, atol=1e-6)
    pose = dRT.RT_transform(d["src_pose"][0][0], np.array([0.9, 0.1, -0.2, 0.05], np.float32),
                            np.array([0.01, -0.02, 0.03], np.float32), np.zeros(3), np.ones(3), "CAMERA", ctx)
    ref = ose3.RT_transform(d["src_pose"][0][0], np.array([0.9, 0.1, -0.2, 0.05], np.float32),
                            np.array([0.01, -0.02, 0.03], np.float32), np.zeros(3), np.ones(3), "CAMERA")
    np.testing.assert_allclose(pose, ref, rtol=1e-6, atol=1e-7)
    fl, vis = d_calc_flow(d["depth_rendered"][0][0, 0], d["src_pose"][0][0], d["pose_tgt"][0], d["K"],
                          d["depth_gt_observed"][0, 0], ctx=ctx)
    rfl, rvis = oflow.calc_flow(d["depth_rendered"][0][0, 0], KT[0], Kinv, d["depth_gt_observed"][0, 0])
    assert np.mean(vis != rvis) < 1e-4
    same = vis == rvis
    np.testing.assert_allclose(fl[same], rfl[same], rtol=1e-4, atol=1e-4)


def test_batch_updater_device(ctx, small_batch):
    """Pose/flow half of batchUpdaterPyMulti.forward on the device vs the oracle composition."""
    from mx_deepim_amd.config import default_config
    from mx_deepim_amd.lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    d = small_batch
    B = d["image_observed"].shape[0]
    cfg = default_config()
    rng = np.random.default_rng(3)
    rot_est = (rng.standard_normal((B, 4)) * 0.05 + [1, 0, 0, 0]).astype(np.float32)
    trans_est = (rng.standard_normal((B, 3)) * 0.02).astype(np.float32)
    upd = batchUpdaterPyMulti(cfg, 480, 640)
    batch = {"src_pose": ctx.array(d["src_pose"][0]), "tgt_pose": ctx.array(d["pose_tgt"]),
             "depth_gt_observed": ctx.array(d["depth_gt_observed"]),
             "next_image_rendered": ctx.array(d["image_rendered"][1]), "next_depth_rendered": ctx.array(d["depth_rendered"][1])}
    new = upd.forward(batch, {"rot_est": ctx.array(rot_est), "trans_est": ctx.array(trans_est)})
    mu, sd = cfg.dataset.trans_means, cfg.dataset.trans_stds
    ref_pose = np.stack([ose3.RT_transform(d["src_pose"][0][b], rot_est[b], trans_est[b], mu, sd, "CAMERA") for b in range(B)])
    np.testing.assert_allclose(new["src_pose"].asnumpy(), ref_pose, rtol=1e-6, atol=1e-7)
    pose32 = new["src_pose"].asnumpy()
    for b in range(B):
        q, t = ose3.calc_RT_delta(pose32[b], d["pose_tgt"][b], mu, sd, "CAMERA", "QUAT")
        np.testing.assert_allclose(new["rot"].asnumpy()[b], q, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(new["trans"].asnumpy()[b], t, rtol=1e-5, atol=1e-6)
    KT = oflow.calc_KT(pose32, d["pose_tgt"], d["K"])
    rf, rv = oflow.gpu_flow(d["depth_rendered"][1], d["depth_gt_observed"], KT, np.linalg.inv(d["K"]).astype(np.float32))
    gv = new["flow_weights"].asnumpy()
    assert np.mean(gv[:, :1] != rv) < 1e-4      # K·T may differ in the last ulp → a threshold tie can flip
    np.testing.assert_array_equal(gv[:, 0], gv[:, 1])
    same = np.broadcast_to(gv[:, :1] == rv, rf.shape)
    np.testing.assert_allclose(new["flow"].asnumpy()[same], rf[same], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(new["mask_rendered"].asnumpy(), (d["depth_rendered"][1] > 0.2).astype(np.float32))
    assert new["image_rendered"] is batch["next_image_rendered"]
