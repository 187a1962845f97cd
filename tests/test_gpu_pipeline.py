"""End-to-end parity of one pose-refinement iteration (config 1 shape: 480x640 pairs) on the GPU
against the CPU oracle: zoom indices/tensors bit-exact, conv stack bit-exact, se3 and pose within
1e-4 relative (north_star tolerance; observed ≈1e-6)."""
import copy

import numpy as np
import pytest

from oracle import pipeline as opipe
from mx_deepim_amd.config import default_config
from mx_deepim_amd.runtime import lib
from mx_deepim_amd.symbols import deepIM_flownet
from mx_deepim_amd import synthetic

pytestmark = pytest.mark.gpu
MEANS_REV = np.ascontiguousarray(synthetic.PIXEL_MEANS[::-1])


def _data(ctx, d, f=0):
    return {"image_observed": ctx.array(d["image_observed"]), "image_rendered": ctx.array(d["image_rendered"][f]),
            "mask_observed": ctx.array(d["mask_observed"]), "mask_rendered": ctx.array(d["mask_rendered"][f]),
            "src_pose": ctx.array(d["src_pose"][f])}


def _np_data(d, f=0):
    return {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][f],
            "mask_observed": d["mask_observed"], "mask_rendered": d["mask_rendered"][f], "src_pose": d["src_pose"][f]}


@pytest.mark.parametrize("nc8", [False, True])
def test_fast_test_iteration_matches_oracle(ctx, small_batch, nc8):
    """nc8 False: NCHW throughout, canonical (ci,ky,kx) chains; True: the default channel-blocked encoder. Either way every
    conv output is one fmaf chain (conv_max_split = 1) in a known order → bit-identical to the oracle run in that order."""
    d = small_batch
    B = d["image_observed"].shape[0]
    cfg = default_config()
    net = deepIM_flownet().get_symbol(cfg)
    params = net.init_weights(cfg, seed=7)
    net.bind(ctx, B, params)
    net.nc8 = nc8
    wino, net.packed_wino = net.packed_wino, {}               # the Winograd layers sum in another order: off for the bit-exact part
    assert "conv3_1" in wino                                  # (bound by default where the layer fills the chip)
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)   # single fmaf chain per output → bit-exact convs
    pose = net.refine_iteration(_data(ctx, d)).asnumpy()
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    ref = opipe.refine_iteration(params, _np_data(d), d["K"], MEANS_REV, cfg.dataset.trans_means, cfg.dataset.trans_stds,
                                 cfg.network.ROT_COORD, nc8=nc8)
    np.testing.assert_array_equal(net.act["zoom_factor"].asnumpy(), ref["zoom_factor"])
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), ref["net_input"])
    for name in ("flow_conv1", "conv3_1", "conv6_1"):
        np.testing.assert_array_equal(net.activation_nchw(name).asnumpy(), ref[name], err_msg=name)
    se3 = net.act["se3"].asnumpy()
    assert np.abs(se3 - ref["se3"]).max() / np.abs(ref["se3"]).max() < 1e-4
    assert np.abs(pose - ref["pose_est"]).max() / np.abs(ref["pose_est"]).max() < 1e-4
    assert np.all(np.isfinite(pose)) and np.abs(ref["conv6_1"]).max() > 1e-3  # activations did not die
    # default policy (auto split-K on the small deep layers, fp32 Winograd on the 3x3 stride-1 layers that fill the chip): same
    # results to fp32 re-association noise
    net.packed_wino = wino
    pose2 = net.refine_iteration(_data(ctx, d)).asnumpy()
    if nc8:
        c3 = net.activation_nchw("conv3_1").asnumpy()
        assert 0 < np.abs(c3 - ref["conv3_1"]).max() <= 1e-5 * np.abs(ref["conv3_1"]).max()     # ran, and within the layer bar
    c2 = net.act["conv6_1"].asnumpy()
    assert np.abs(c2 - ref["conv6_1"]).max() <= 1e-5 * np.abs(ref["conv6_1"]).max()
    assert np.abs(pose2 - ref["pose_est"]).max() / np.abs(ref["pose_est"]).max() < 1e-4


@pytest.mark.parametrize("nc8", [False, True])
def test_heads_iteration_matches_oracle(ctx, small_batch, nc8):
    d = small_batch
    B = 1
    d1 = {k: (v[:, :B] if k in ("image_rendered", "mask_rendered", "depth_rendered", "src_pose") else (v[:B] if k != "K" else v))
          for k, v in d.items()}
    cfg = default_config()
    cfg.TEST.FAST_TEST = False
    net = deepIM_flownet().get_symbol(cfg)
    assert net.with_mask_head and net.with_flow_head
    params = net.init_weights(cfg, seed=8)
    net.bind(ctx, B, params)
    net.nc8 = nc8
    net.packed_wino = {}        # bit-exact comparison: every layer on the direct kernels (none is bound at B = 1 anyway)
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
    out = net.forward(_data(ctx, d1))
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    ref = opipe.refine_iteration(params, _np_data(d1), d["K"], MEANS_REV, cfg.dataset.trans_means,
                                 cfg.dataset.trans_stds, cfg.network.ROT_COORD, heads=True,
                                 normalize_flow=cfg.dataset.NORMALIZE_FLOW, nc8=nc8)
    for name in ("flow6", "Concat2", "flow5", "Concat3", "mask_lowres", "flow_lowres"):
        np.testing.assert_array_equal(net.act[name].asnumpy(), ref[name], err_msg=name)
    np.testing.assert_array_equal(net.act["mask_logits"].asnumpy(), ref["mask_logits"])
    np.testing.assert_array_equal(net.act["zoom_flow_est"].asnumpy(), ref["zoom_flow_est"])
    flow = out["flow_est_crop"].asnumpy()
    assert np.abs(flow - ref["flow_est"]).max() <= 1e-4 * max(1.0, np.abs(ref["flow_est"]).max())
    mism = np.mean(out["mask_observed_pred"].asnumpy() != ref["mask_observed_pred"])
    assert mism < 1e-4, mism  # sigmoid ulp differences can flip a pixel sitting on the 0.2 threshold
    assert np.abs(out["se3"].asnumpy() - ref["se3"]).max() / np.abs(ref["se3"]).max() < 1e-4


def test_four_iteration_loop_runs_and_is_deterministic(ctx, small_batch):
    d = small_batch
    B = d["image_observed"].shape[0]
    cfg = default_config()
    net = deepIM_flownet().get_symbol(cfg)
    net.bind(ctx, B, net.init_weights(cfg, seed=7))

    def run():
        data = _data(ctx, d, 0)
        poses = []
        for it in range(4):
            f = min(it, d["image_rendered"].shape[0] - 1)
            data["image_rendered"] = ctx.array(d["image_rendered"][f])
            data["mask_rendered"] = ctx.array(d["mask_rendered"][f])
            pose = net.refine_iteration(data).copy()
            data["src_pose"] = pose
            poses.append(pose.asnumpy())
        return np.stack(poses)

    a, b = run(), run()
    np.testing.assert_array_equal(a, b)
    assert np.all(np.isfinite(a))


def test_hipgraph_replay_matches_eager(ctx, small_batch):
    """One refinement iteration captured into a hipGraph and replayed on refreshed inputs == eager launches."""
    d = small_batch
    B = d["image_observed"].shape[0]
    cfg = default_config()
    net = deepIM_flownet().get_symbol(cfg)
    net.bind(ctx, B, net.init_weights(cfg, seed=7))
    data = _data(ctx, d, 0)
    pose_out = ctx.empty((B, 3, 4))
    eager0 = net.refine_iteration(data, pose_out).asnumpy()          # eager first: first-call work happens here
    gid = net.capture_iteration(data, pose_out)
    net.replay(gid)
    np.testing.assert_array_equal(pose_out.asnumpy(), eager0)
    # refresh the CONTENTS of the captured buffers with frame 1 and replay
    data["image_rendered"].copyfrom(d["image_rendered"][1])
    data["mask_rendered"].copyfrom(d["mask_rendered"][1])
    data["src_pose"].copyfrom(d["src_pose"][1])
    net.replay(gid)
    got = pose_out.asnumpy()
    eager1 = net.refine_iteration(data).asnumpy()
    np.testing.assert_array_equal(got, eager1)
    assert not np.array_equal(eager0, eager1)


def test_opt_in_conv1_from_channel_blocked_net_input(ctx, small_batch):
    """`net.conv1_nc8 = True`: the zoom front end writes (B,H,W,8) records and conv1 runs on the 64x256-tile NC8 kernel.
    Net input bit-exact (through the lazily converted NCHW view), every encoder layer bit-exact against the oracle in that
    summation order, pose within 1e-4. (Measured slower than the default — see deepIM_flownet._conv1_from_nc8 — hence opt-in.)"""
    from oracle import pipeline as opipe
    from mx_deepim_amd import synthetic
    from mx_deepim_amd.config import default_config
    from mx_deepim_amd.runtime import lib
    from mx_deepim_amd.symbols import deepIM_flownet
    d = small_batch
    cfg = default_config()
    cfg.network.WINOGRAD_CONV = False      # bit-exact comparison below: every layer on the direct kernels' fmaf chains
    net = deepIM_flownet().get_symbol(cfg)
    params = net.init_weights(cfg, seed=5)
    net.bind(ctx, 2, params)
    assert not net.packed_wino
    net.conv1_nc8 = True
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    data = {k: ctx.array(d[k]) for k in ("image_observed", "mask_observed")}
    data.update({k: ctx.array(d[k][0]) for k in ("image_rendered", "mask_rendered", "src_pose")})
    lib.deepim_set_option(ctx.handle, b"conv_force_plan", 1)       # one chain per output: bit-exact against the oracle order
    try:
        pose = net.refine_iteration(data).asnumpy()
    finally:
        lib.deepim_set_option(ctx.handle, b"conv_force_plan", 0)
    assert net._input_live_nc8
    host = {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][0], "mask_observed": d["mask_observed"],
            "mask_rendered": d["mask_rendered"][0], "src_pose": d["src_pose"][0]}
    ref = opipe.refine_iteration(params, host, d["K"], np.ascontiguousarray(synthetic.PIXEL_MEANS[::-1]), cfg.dataset.trans_means,
                                 cfg.dataset.trans_stds, cfg.network.ROT_COORD, nc8=True, conv1_nc8=True)
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), ref["net_input"])
    np.testing.assert_array_equal(net.activation_nchw("flow_conv1").asnumpy(), ref["flow_conv1"])
    np.testing.assert_array_equal(net.act["conv6_1"].asnumpy(), ref["conv6_1"])
    assert np.abs(pose - ref["pose_est"]).max() / np.abs(ref["pose_est"]).max() < 1e-4


@pytest.mark.parametrize("B", [1, 3, 5])
def test_fused_pose_tail_is_bit_identical_to_the_separate_launches(ctx, B):
    """deepim_pose_tail_forward (fc7 → rot / trans + inverse ZoomTrans → se3 → RT_transform in ONE launch, round 4) against
    deepim_fc_forward + deepim_pose_head_forward + deepim_rt_transform on the same inputs: fc7, se3 and the refined poses equal
    bit for bit in all four rot_coords, also when the poses are updated in place (pose_out = src_pose, as the loop does)."""
    import ctypes
    rng = np.random.default_rng(40 + B)
    cf = ctypes.c_float
    fc6 = rng.standard_normal((B, 256)).astype(np.float32)
    w7 = (rng.standard_normal((256, 256)) / 16).astype(np.float32)
    b7 = (rng.standard_normal(256) * 0.1).astype(np.float32)
    w_rot = (rng.standard_normal((4, 256)) * 0.05).astype(np.float32)
    b_rot = np.array([1.0, 0.01, -0.02, 0.03], np.float32)
    w_tr = (rng.standard_normal((3, 256)) * 0.01).astype(np.float32)
    b_tr = (rng.standard_normal(3) * 0.01).astype(np.float32)
    zf = np.concatenate([rng.uniform(0.3, 0.9, (B, 1)).repeat(2, 1), rng.uniform(-0.2, 0.2, (B, 2))], 1).astype(np.float32)
    poses = np.stack([synthetic.sample_pose_pair(rng)[1] for _ in range(B)]).astype(np.float32)
    mu, sd = np.array([0.01, -0.02, 0.03], np.float32), np.array([0.9, 1.1, 1.05], np.float32)
    d = {k: ctx.array(v) for k, v in dict(fc6=fc6, w7=w7, b7=b7, w_rot=w_rot, b_rot=b_rot, w_tr=w_tr, b_tr=b_tr, zf=zf).items()}
    h = ctx.handle
    for rc in range(4):
        src = ctx.array(poses)
        f7a, se3a, posea = ctx.empty((B, 256)), ctx.empty((B, 7)), ctx.empty((B, 3, 4))
        lib.deepim_fc_forward(h, f7a, d["fc6"], d["w7"], d["b7"], B, 256, 256, cf(0.1))
        lib.deepim_pose_head_forward(h, se3a, f7a, d["w_rot"], d["b_rot"], d["w_tr"], d["b_tr"], d["zf"], B, 256)
        lib.deepim_rt_transform(h, posea, None, src, se3a, mu, sd, rc, B)
        f7b, se3b, poseb = ctx.empty((B, 256)), ctx.empty((B, 7)), ctx.empty((B, 3, 4))
        lib.deepim_pose_tail_forward(h, f7b, se3b, poseb, d["fc6"], d["w7"], d["b7"], d["w_rot"], d["b_rot"], d["w_tr"], d["b_tr"],
                                     d["zf"], src, mu, sd, rc, B, 256, cf(0.1))
        np.testing.assert_array_equal(f7b.asnumpy(), f7a.asnumpy())
        np.testing.assert_array_equal(se3b.asnumpy(), se3a.asnumpy())
        np.testing.assert_array_equal(poseb.asnumpy(), posea.asnumpy())
        assert np.abs(poseb.asnumpy() - poses).max() > 1e-4
        lib.deepim_pose_tail_forward(h, f7b, se3b, src, d["fc6"], d["w7"], d["b7"], d["w_rot"], d["b_rot"], d["w_tr"], d["b_tr"],
                                     d["zf"], src, mu, sd, rc, B, 256, cf(0.1))            # in place
        np.testing.assert_array_equal(src.asnumpy(), posea.asnumpy())
    with pytest.raises(RuntimeError):
        lib.deepim_pose_tail_forward(h, f7b, se3b, poseb, d["fc6"], d["w7"], d["b7"], d["w_rot"], d["b_rot"], d["w_tr"], d["b_tr"],
                                     d["zf"], src, mu, sd, 0, B, 128, cf(0.1))


def test_euler_test_graph_iteration(ctx, small_batch):
    """network.ROT_TYPE = "EULER" (deepIM_flownet.py:715, :791-793; tester.py:391-398): a 3-output rot head, se3 = [euler | trans]
    (B, 6), RT_transform's Euler branch. The head against numpy on the GPU's own fc7, the pose update against the oracle's
    RT_transform path that tests/golden/se3_extra_golden.npz pins (euler2mat + R_transform / T_transform)."""
    from oracle import se3 as ose3, zoom as ozoom
    from mx_deepim_amd.lib.pair_matching import RT_transform as RT
    d = small_batch
    B = d["image_observed"].shape[0]
    cfg = default_config()
    cfg.network.ROT_TYPE = "EULER"
    net = deepIM_flownet().get_symbol(cfg)
    assert net.arg_shape_dict()["rot_weight"] == (3, 256) and net.arg_shape_dict()["rot_bias"] == (3,)
    params = net.init_weights(cfg, seed=7)
    assert not params["rot_weight"].any()                       # :791-792 zero-initialised Euler head
    rng = np.random.default_rng(3)
    params["rot_weight"] = (0.02 * rng.standard_normal((3, 256))).astype(np.float32)     # a head that actually rotates
    net.bind(ctx, B, params)
    data = _data(ctx, d)
    pose = net.refine_iteration(data).asnumpy()
    se3 = net.act["se3"].asnumpy()
    assert se3.shape == (B, 6)
    fc7 = net.act["fc7"].asnumpy().astype(np.float64)
    rot = fc7 @ params["rot_weight"].astype(np.float64).T + params["rot_bias"]
    ztr = (fc7 @ params["trans_weight"].astype(np.float64).T + params["trans_bias"]).astype(np.float32)
    tr = ozoom.zoom_trans(net.act["zoom_factor"].asnumpy(), ztr, b_inv_zoom=True)
    assert np.abs(se3[:, :3] - rot).max() <= 1e-5 * max(1.0, np.abs(rot).max())
    assert np.abs(se3[:, 3:] - tr).max() <= 1e-5 * max(1.0, np.abs(tr).max())
    assert np.abs(se3[:, :3]).max() > 1e-4
    # pose update = RT_transform.py:138-151 with euler2mat (axes sxyz): the host mirror's own numpy path is pinned by the golden file
    src = d["src_pose"][0]
    for b in range(B):
        want = np.zeros((3, 4))
        want[:, :3] = ose3.R_transform(src[b][:, :3].astype(np.float64), RT.euler2mat(*se3[b, :3].astype(np.float64)), cfg.network.ROT_COORD)
        want[:, 3] = ose3.T_transform(src[b][:, 3].astype(np.float64), se3[b, 3:].astype(np.float64), cfg.dataset.trans_means,
                                      cfg.dataset.trans_stds, cfg.network.ROT_COORD)
        assert np.abs(pose[b] - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
    # the separate launches (forward() + pose_update) give the same poses as refine_iteration
    net.forward(data)
    np.testing.assert_array_equal(net.pose_update(data["src_pose"]).asnumpy(), pose)
    with pytest.raises(Exception, match="rot_type"):
        cfg.network.ROT_TYPE = "AXIS"
        deepIM_flownet().get_symbol(cfg)


@pytest.mark.parametrize("style", ["deepim", "flownet"])
def test_checkpoint_import_reaches_the_bound_network(ctx, small_batch, tmp_path, style):
    """SURVEY 8(f2), on the device: a `.params` file (`arg:` / `aux:` keys, `_test` / `_i2r` suffixes — lib/utils/load_model.py:21-30,
    :61-67) → load_param → init_weights (6-channel FlowNet conv1 zero-padded to the 8-channel graph, deepIM_flownet.py:759-773; what the
    file lacks is initialised) → bind. Every bound device tensor equals the file; the packed forms are checked through what they
    compute — the Winograd operand of conv3_1 element by element, and one refinement iteration against the oracle run on the FILE's
    weights. (The container layout itself is restated from MXNet 1.2 and round-trips through this repo's writer only: unpinned until an
    MXNet-written file is available — DESIGN.md section 4.)"""
    from mx_deepim_amd.lib.utils.load_model import load_param, save_checkpoint
    d = small_batch
    B = d["image_observed"].shape[0]
    cfg = default_config()
    net = deepIM_flownet().get_symbol(cfg)
    shapes = net.arg_shape_dict()
    full = deepIM_flownet().get_symbol(cfg).init_weights(cfg, seed=41)
    if style == "deepim":      # a DeepIM checkpoint: every parameter of the graph, some under the suffixes older files carry
        ckpt = {(k.replace("_weight", "_weight_test") if k in ("fc6_weight", "rot_weight") else
                 k.replace("_bias", "_i2r_bias") if k == "conv4_1_bias" else k): v for k, v in full.items()}
        aux = {"bn_dummy_moving_mean": np.zeros(4, np.float32)}
    else:                       # a FlowNetS checkpoint: the encoder only, conv1 on 6 RGB-pair channels
        ckpt = {k: v for k, v in full.items() if k.startswith(("flow_conv1", "conv"))}
        ckpt["flow_conv1_weight"] = np.ascontiguousarray(full["flow_conv1_weight"][:, :6])
        aux = {}
    prefix = str(tmp_path / style)
    save_checkpoint(prefix, 3, ckpt, aux)
    arg, aux_l = load_param(prefix, 3, process=True)
    assert set(aux_l) == set(aux) and not any("_test" in k or "_i2r" in k for k in arg)
    params = net.init_weights(cfg, arg_params=arg, seed=5)
    assert set(params) == set(shapes)
    net.bind(ctx, B, params)
    for name in shapes:
        got = net.params[name].asnumpy()
        if name == "flow_conv1_weight" and style == "flownet":
            np.testing.assert_array_equal(got[:, :6], full[name][:, :6])
            assert not got[:, 6:].any()                          # the mask channels start from zero weights
        elif style == "deepim" or name in ckpt:
            np.testing.assert_array_equal(got, full[name], err_msg=name)
        else:
            assert np.isfinite(got).all() and got.shape == tuple(shapes[name])
    # the Winograd operand of conv3_1 is G g G^T of the FILE's weights (positions nu = 3 negated), element by element
    assert "conv3_1" in net.packed_wino
    w = full["conv3_1_weight"]
    pk = net.packed_wino["conv3_1"].asnumpy().reshape(w.shape[0] // 32, w.shape[1] // 8, 16, 2, 32, 4)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
    U = np.einsum("xa,ocab,nb->ocxn", G, w.astype(np.float64), G).astype(np.float32)
    U[..., 3] = -U[..., 3]
    for mb, c8, h, s_ in ((0, 0, 0, 0), (3, 17, 1, 2), (7, 31, 1, 3)):
        np.testing.assert_array_equal(pk[mb, c8, :, h, :, s_], U[mb * 32:(mb + 1) * 32, c8 * 8 + 4 * h + s_].reshape(32, 16).T)
    # ... and the network computes with them: one iteration against the oracle on the merged host parameters
    host = {k: net.params[k].asnumpy() for k in shapes}
    pose = net.refine_iteration(_data(ctx, d)).asnumpy()
    ref = opipe.refine_iteration(host, _np_data(d), d["K"], MEANS_REV, cfg.dataset.trans_means, cfg.dataset.trans_stds,
                                 cfg.network.ROT_COORD, nc8=True)
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), ref["net_input"])
    c6 = net.act["conv6_1"].asnumpy()
    assert np.abs(c6 - ref["conv6_1"]).max() <= 1e-5 * np.abs(ref["conv6_1"]).max() and np.abs(ref["conv6_1"]).max() > 1e-3
    assert np.abs(pose - ref["pose_est"]).max() / np.abs(ref["pose_est"]).max() < 1e-4
