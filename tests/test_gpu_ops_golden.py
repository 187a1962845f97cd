"""The HIP kernels of S5/S6 (Transform3D fwd/bwd), F3 (FlowUpdater) and GroupPicker checked DIRECTLY against
tests/golden/ops_golden.npz — outputs of the reference's own deepim/operator_py/{transform3d,flow_updater,
group_picker}.py run unmodified over the fake-mxnet rig (tests/golden/make_ops_golden.py) — through the C ABI and
through the CustomOp mirrors.  Target reading: NumPy-1.x promotion ("legacy"); the third-party reductions inside MXNet
(batch_dot / sum) exist in two readings (float32 sequential | float64 accumulation), the kernels must agree with both
inside the bar.

Bars: Transform3D forward <= 1e-6 of the output maximum (observed printed), identity-rotation rows selected exactly as
the reference's |Nq-1| < 1e-2 gate does; backward <= 1e-4 of the gradient maximum (north_star), zero-gradient rows
exactly as the |Nq-1| < 1e-4 gate does; FlowUpdater flow and weights BIT-EXACT; GroupPicker exact.
"""
import os

import numpy as np
import pytest

from mx_deepim_amd import mx
from mx_deepim_amd import operator_py  # noqa: F401  (registers the ops)
from mx_deepim_amd.runtime import lib

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ops_golden.npz"))
COORDS = {"MODEL": 0, "CAMERA": 1, "CAMERA_NEW": 2, "NAIVE": 3}


def _t3d(ctx):
    names = ("points", "rotation", "translation", "pose_src", "out_grad")
    host = {k: G["t3d_" + k] for k in names}
    dev = {k: ctx.array(v) for k, v in host.items()}
    return host, dev, np.ascontiguousarray(G["t3d_T_means"]), np.ascontiguousarray(G["t3d_T_stds"])


@pytest.mark.parametrize("coord", list(COORDS))
def test_transform3d_forward_vs_reference_run(ctx, coord):
    host, dev, mu, sd = _t3d(ctx)
    B, _, N = host["points"].shape
    out = ctx.empty((B, 3, N))
    lib.deepim_transform3d_forward(ctx.handle, out, dev["points"], dev["rotation"], dev["translation"], dev["pose_src"],
                                   mu, sd, COORDS[coord], B, N)
    y = out.asnumpy()
    ref = G["t3d_out_%s_legacy_seq" % coord]
    err = np.abs(y.astype(np.float64) - ref).max() / np.abs(ref).max()
    err64 = np.abs(y[:, :, ::16].astype(np.float64) - G["t3d_out_%s_legacy_f64" % coord]).max() / np.abs(ref).max()
    print("Transform3D fwd %s: max rel err vs reference run %.2e (seq reading), %.2e (f64 reading)" % (coord, err, err64))
    assert err <= 1e-6 and err64 <= 1e-6
    # the forward gate: samples with |Nq - 1| >= 1e-2 use the identity rotation — a wrong gate would be an O(0.1) error there
    off = G["t3d_nq_offsets"]
    for b in np.nonzero(np.abs(off) >= 1e-2)[0]:
        assert np.abs(y[b] - ref[b]).max() <= 1e-6 * np.abs(ref).max()


@pytest.mark.parametrize("coord", list(COORDS))
def test_transform3d_backward_vs_reference_run(ctx, coord):
    host, dev, mu, sd = _t3d(ctx)
    B, _, N = host["points"].shape
    gq, gt = ctx.empty((B, 4)), ctx.empty((B, 3))
    lib.deepim_transform3d_backward(ctx.handle, gq, gt, dev["out_grad"], dev["points"], dev["rotation"], dev["translation"],
                                    dev["pose_src"], mu, sd, COORDS[coord], B, N)
    dq, dt = gq.asnumpy(), gt.asnumpy()
    for acc in ("seq", "f64"):
        rq, rt = G["t3d_drot_%s_legacy_%s" % (coord, acc)], G["t3d_dtrans_%s_legacy_%s" % (coord, acc)]
        eq = np.abs(dq - rq).max() / np.abs(rq).max()
        et = np.abs(dt - rt).max() / np.abs(rt).max()
        print("Transform3D bwd %s vs reference run (%s reading): d_rotation %.2e, d_translation %.2e" % (coord, acc, eq, et))
        assert eq <= 1e-4 and et <= 1e-4
    # the backward gate (|Nq - 1| < 1e-4): zero rows exactly where the reference returns zeros, nowhere else
    zero_ref = ~G["t3d_drot_%s_legacy_seq" % coord].any(axis=1)
    assert np.array_equal(~dq.any(axis=1), zero_ref)
    assert zero_ref.sum() == 6 and (~zero_ref).sum() == 3


def test_transform3d_operator_mirror_vs_reference_run(ctx):
    """Same fixture through mx.nd.Custom-style forward/backward of operator_py/transform3d.py (string attrs)."""
    host, dev, mu, sd = _t3d(ctx)
    prop = mx.operator.get_registered("Transform3D")(T_means=str(mu), T_stds=str(sd), rot_coord="CAMERA", b_project_2d="False")
    assert prop.list_arguments() == ["point_cloud", "rotation", "translation", "pose_src"]
    opr = prop.create_operator(ctx, None, None)
    in_data = [dev["points"], dev["rotation"], dev["translation"], dev["pose_src"]]
    out = [ctx.empty(host["points"].shape)]
    opr.forward(True, ["write"], in_data, out, [])
    ref = G["t3d_out_CAMERA_legacy_seq"]
    assert np.abs(out[0].asnumpy() - ref).max() <= 1e-6 * np.abs(ref).max()
    in_grad = [ctx.array(np.full_like(host[k], 7.0)) for k in ("points", "rotation", "translation", "pose_src")]
    opr.backward(["write"] * 4, [dev["out_grad"]], in_data, out, in_grad, [])
    assert not in_grad[0].asnumpy().any() and not in_grad[3].asnumpy().any()
    rq, rt = G["t3d_drot_CAMERA_legacy_seq"], G["t3d_dtrans_CAMERA_legacy_seq"]
    assert np.abs(in_grad[1].asnumpy() - rq).max() <= 1e-4 * np.abs(rq).max()
    assert np.abs(in_grad[2].asnumpy() - rt).max() <= 1e-4 * np.abs(rt).max()


def _unpack(flow_i16, wbits, shape):
    n = int(np.prod(shape))
    return flow_i16.astype(np.float32), np.unpackbits(wbits)[:n].reshape(shape).astype(np.float32)


@pytest.mark.parametrize("wh", [0, 1])
def test_flow_updater_small_vs_reference_run(ctx, wh):
    a = [np.ascontiguousarray(G["fu_small_" + k]) for k in ("depth_src", "depth_tgt", "pose_src", "pose_tgt", "K")]
    B, _, H, W = a[0].shape
    flow, wts = ctx.empty((B, 2, H, W)), ctx.empty((B, 2, H, W))
    lib.deepim_flow_updater_forward(ctx.handle, flow, wts, ctx.array(a[0]), ctx.array(a[1]), ctx.array(a[2]), ctx.array(a[3]),
                                    a[4], 3e-3, wh, B, H, W)
    rf, rw = _unpack(G["fu_small_flow_legacy_seq_wh%d" % wh], G["fu_small_wbits_legacy_seq_wh%d" % wh], (B, 2, H, W))
    assert rw.sum() > 500
    np.testing.assert_array_equal(wts.asnumpy(), rw)
    np.testing.assert_array_equal(flow.asnumpy(), rf)


def test_flow_updater_480x640_vs_reference_run(ctx, small_batch):
    d = small_batch
    np.testing.assert_array_equal(d["src_pose"][0], G["fu_full_pose_src"])
    B, _, H, W = d["depth_rendered"][0].shape
    flow, wts = ctx.empty((B, 2, H, W)), ctx.empty((B, 2, H, W))
    lib.deepim_flow_updater_forward(ctx.handle, flow, wts, ctx.array(d["depth_rendered"][0]), ctx.array(d["depth_gt_observed"]),
                                    ctx.array(d["src_pose"][0]), ctx.array(d["pose_tgt"]), np.ascontiguousarray(d["K"]),
                                    3e-3, 0, B, H, W)
    rf, rw = _unpack(G["fu_full_flow_legacy_seq"], G["fu_full_wbits_legacy_seq"], (B, 2, H, W))
    assert rw.sum() > 5000
    np.testing.assert_array_equal(wts.asnumpy(), rw)
    np.testing.assert_array_equal(flow.asnumpy(), rf)
    # and through the CustomOp mirror, string attrs as MXNet hands them over
    res = mx.nd.Custom(ctx.array(d["depth_rendered"][0]), ctx.array(d["depth_gt_observed"]), ctx.array(d["src_pose"][0]),
                       ctx.array(d["pose_tgt"]), op_type="FlowUpdater", K=str(d["K"].flatten()), thresh="0.003",
                       batch_size=str(B), height=str(H), width=str(W), wh_rep="False")
    np.testing.assert_array_equal(res[0].asnumpy(), rf)
    np.testing.assert_array_equal(res[1].asnumpy(), rw)


def test_group_picker_vs_reference_run(ctx):
    x, idx = G["gp_x"], G["gp_idx"]                     # (5, 12, 3, 4) with 4 groups: trailing axes fold into the channel run
    B, C = x.shape[0], int(np.prod(x.shape[1:]))
    out = ctx.empty(G["gp_out"].shape)
    lib.deepim_group_picker_forward(ctx.handle, out, ctx.array(x), ctx.array(idx.reshape(-1)), 4, B, C)
    np.testing.assert_array_equal(out.asnumpy(), G["gp_out"])
    gin = ctx.empty(x.shape)
    lib.deepim_group_picker_backward(ctx.handle, gin, ctx.array(G["gp_out_grad"]), ctx.array(idx.reshape(-1)), 4, B, C)
    np.testing.assert_array_equal(gin.asnumpy(), G["gp_dx"])
    # the CustomOp mirror on both fixture shapes
    for pre, gn in (("gp", 4), ("gp2", 2)):
        xi, ii = G[pre + "_x"], G[pre + "_idx"]
        prop = mx.operator.get_registered("GroupPicker")(group_num=str(gn))
        opr = prop.create_operator(ctx, None, None)
        in_data = [ctx.array(xi), ctx.array(ii)]
        o = [ctx.empty(G[pre + "_out"].shape)]
        opr.forward(True, ["write"], in_data, o, [])
        np.testing.assert_array_equal(o[0].asnumpy(), G[pre + "_out"])
        ig = [ctx.array(np.full_like(xi, 7.0)), ctx.array(np.full_like(ii, 7.0))]
        opr.backward(["write", "write"], [ctx.array(G[pre + "_out_grad"])], in_data, o, ig, [])
        np.testing.assert_array_equal(ig[0].asnumpy(), G[pre + "_dx"])
        assert not ig[1].asnumpy().any()
