"""Parity tests on BASELINE.json's configs 3, 4 and 5 AS WRITTEN, GPU (C ABI, the default kernels that
bench.py times) against the CPU oracle:

  config 3  13-object LINEMOD batch sharded by object: 13 meshes, a mixed `class_index` batch at the per-GPU share
            (B = 4), two closed-loop refinement iterations (refine → re-render each object's own mesh → mask update →
            refine).  Bars: zoom crop indices / net input bit-exact; refined pose ≤ 1e-4 relative; re-rendered depth,
            rendered mask and box_rendered rectangle bit-exact against the oracle renderer at the same pose.
  config 4  Occlusion-LINEMOD: an occluder rectangle painted over the observed frame, B = 2, full test graph (decoder +
            mask / flow heads).  Bars: flow ≤ 1e-4, mask flips < 1e-4 of the pixels, se3 ≤ 1e-4.
  configs 3 and 4 run twice: on the default fp32 kernels and with network.X3_CONV (split-fp16 convs) — the same bars.
  config 5  ModelNet RGB-D: INPUT_DEPTH=True (C_in = 10 → padded to 16 in the fp16 NHWC layout), conv stack on the
            fp16 matrix cores.  Bars (fp16 cannot meet 1e-4): conv6_1 ≤ 2e-3, se3 ≤ 1e-3, pose ≤ 1e-4 against the oracle's
            fp16 emulation (fp16-rounded operands/outputs, fp32 accumulate).
"""
import numpy as np
import pytest

from oracle import flow as oflow
from oracle import pipeline as opipe
from oracle import render as orender
from oracle import zoom as oz
from mx_deepim_amd import synthetic
from mx_deepim_amd.config import default_config
from mx_deepim_amd.lib.pair_matching.batch_updater_py_multi import update_test_batch
from mx_deepim_amd.lib.render_glumpy.render_py_multi import Render_Py
from mx_deepim_amd.runtime import lib
from mx_deepim_amd.symbols import deepIM_flownet

pytestmark = pytest.mark.gpu
MEANS_REV = np.ascontiguousarray(synthetic.PIXEL_MEANS[::-1])
H, W = 480, 640
LM_CLASSES = ["ape", "benchvise", "camera", "can", "cat", "driller", "duck", "eggbox", "glue", "holepuncher", "iron",
              "lamp", "phone"]


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())


def _meshes13():
    rng = np.random.default_rng(13)
    meshes = {}
    for i, c in enumerate(LM_CLASSES):
        m = synthetic.ellipsoid_mesh(np.array([0.05, 0.04, 0.035]) * rng.uniform(0.7, 1.4, 3), 12, 24)
        m.pop("uv")
        m["colors"] = np.floor(np.clip(m["colors"] * rng.uniform(0.4, 1.0, 3) + 10 * i, 0, 255)).astype(np.float32)
        meshes[c] = m
    return meshes


def _render_frame(mesh, pose, K):
    img, dep = orender.render(mesh["vertices"], mesh["colors"], mesh["faces"], pose, K, H, W, pixel_means=MEANS_REV)
    return img, dep


@pytest.mark.parametrize("x3", [False, True], ids=["fp32", "split_fp16_x3"])
def test_config3_thirteen_objects_mixed_batch_two_closed_loop_iterations(ctx, x3):
    """x3: the same bars with the encoder on the split-fp16 kernels (fp32-grade, not bit-exact: the pose bar is 1e-4 either way)."""
    cfg = default_config()
    cfg.network.X3_CONV = x3
    K = cfg.dataset.INTRINSIC_MATRIX
    meshes = _meshes13()
    class_index = np.array([3, 3, 11, 7])            # a run of two 'can's, then 'lamp', then 'eggbox'
    B = len(class_index)
    rng = np.random.default_rng(303)
    img_o, img_r, mask_r, mask_o, src = [], [], [], [], []
    for b in range(B):
        tgt, s = synthetic.sample_pose_pair(rng, K, H, W)
        mesh = meshes[LM_CLASSES[class_index[b]]]
        io, do = _render_frame(mesh, tgt, K)
        bg = (np.floor(rng.uniform(0, 255, (3, H, W))).astype(np.float32) - MEANS_REV.reshape(3, 1, 1)).astype(np.float32)
        img_o.append(np.where(do[None] > 0, io, bg))
        ir, dr = _render_frame(mesh, s, K)
        img_r.append(ir)
        m = (dr > 0.2).astype(np.float32)
        assert m.any()
        mask_r.append(m[None])
        mask_o.append(oflow.mask_box(m)[None])
        src.append(s)
    d = {"image_observed": np.stack(img_o).astype(np.float32), "image_rendered": np.stack(img_r).astype(np.float32),
         "mask_rendered": np.stack(mask_r), "mask_observed": np.stack(mask_o), "src_pose": np.stack(src).astype(np.float32)}

    net = deepIM_flownet().get_symbol(cfg)
    params = net.init_weights(cfg, seed=33)
    params["trans_weight"] = params["trans_weight"] * np.float32(0.02)     # keep the object in frame (as bench.py does)
    params["trans_bias"] = params["trans_bias"] * np.float32(0.02)
    net.bind(ctx, B, params)
    rm = Render_Py("unused", LM_CLASSES, K, W, H, meshes=meshes, ctx=ctx, pixel_means=MEANS_REV.copy())
    args = (K, MEANS_REV, cfg.dataset.trans_means, cfg.dataset.trans_stds, cfg.network.ROT_COORD)

    # ---- iteration 1
    data = {k: ctx.array(v) for k, v in d.items()}
    pose1 = net.refine_iteration(data).copy()
    ref1 = opipe.refine_iteration(params, d, *args, nc8=True)
    np.testing.assert_array_equal(net.act["zoom_factor"].asnumpy(), ref1["zoom_factor"])
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), ref1["net_input"])
    idx = ctx.empty((B, 2, H, W), dtype=np.int32)
    lib.deepim_zoom_indices(ctx.handle, net.act["zoom_factor"], idx, B, H, W)
    np.testing.assert_array_equal(idx.asnumpy(), oz.sample_indices(ref1["zoom_factor"], H, W))
    p1 = pose1.asnumpy()
    assert rel(net.act["se3"].asnumpy(), ref1["se3"]) < 1e-4
    assert rel(p1, ref1["pose_est"]) < 1e-4

    # ---- re-render of every sample with ITS OWN class mesh + mask update (tester.py:420-455), checked at the GPU's pose
    data2 = update_test_batch(cfg, data, rm, pose1, class_index=class_index)
    d2 = dict(d)
    d2["src_pose"] = p1
    ir2, mr2, mo2 = [], [], []
    for b in range(B):
        ri, rd = _render_frame(meshes[LM_CLASSES[class_index[b]]], p1[b], K)
        ir2.append(ri)
        mr2.append((rd > 0.2).astype(np.float32)[None])
        mo2.append(oflow.mask_box(mr2[-1][0])[None])
    d2["image_rendered"], d2["mask_rendered"], d2["mask_observed"] = np.stack(ir2).astype(np.float32), np.stack(mr2), np.stack(mo2)
    np.testing.assert_array_equal(data2["mask_rendered"].asnumpy(), d2["mask_rendered"])
    np.testing.assert_array_equal(data2["mask_observed"].asnumpy(), d2["mask_observed"])
    np.testing.assert_allclose(data2["image_rendered"].asnumpy(), d2["image_rendered"], atol=1e-3)
    assert len({tuple(np.argwhere(m[0]).min(0)) for m in d2["mask_rendered"]}) > 1   # the samples really differ

    # ---- iteration 2 on the re-rendered frames (oracle fed the frames the GPU drew, so both see identical inputs)
    pose2 = net.refine_iteration(data2).asnumpy()
    d2["image_rendered"] = data2["image_rendered"].asnumpy()
    ref2 = opipe.refine_iteration(params, d2, *args, nc8=True)
    np.testing.assert_array_equal(net.act["zoom_factor"].asnumpy(), ref2["zoom_factor"])
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), ref2["net_input"])
    assert rel(net.act["se3"].asnumpy(), ref2["se3"]) < 1e-4
    assert rel(pose2, ref2["pose_est"]) < 1e-4
    assert rel(pose2, p1) > 1e-4                                                      # the second iteration moved the pose


@pytest.mark.parametrize("x3", [False, True], ids=["fp32", "split_fp16_x3"])
def test_config4_occluded_batch_with_decoder_and_heads(ctx, x3):
    d = synthetic.make_batch(2, seed=404, n_frames=1, occlude=True)
    B = 2
    cfg = default_config()
    cfg.network.X3_CONV = x3
    cfg.TEST.FAST_TEST = False
    net = deepIM_flownet().get_symbol(cfg)
    assert net.with_mask_head and net.with_flow_head
    params = net.init_weights(cfg, seed=44)
    net.bind(ctx, B, params)
    npd = {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][0], "mask_observed": d["mask_observed"],
           "mask_rendered": d["mask_rendered"][0], "src_pose": d["src_pose"][0]}
    # the occluder really covers part of an observed object
    clean = synthetic.make_batch(2, seed=404, n_frames=1, occlude=False)
    assert (clean["image_observed"] != d["image_observed"]).any()
    out = net.forward({k: ctx.array(v) for k, v in npd.items()})
    ref = opipe.refine_iteration(params, npd, d["K"], MEANS_REV, cfg.dataset.trans_means, cfg.dataset.trans_stds,
                                 cfg.network.ROT_COORD, heads=True, normalize_flow=cfg.dataset.NORMALIZE_FLOW, nc8=True)
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), ref["net_input"])
    for name in ("Concat2", "Concat3", "mask_logits", "zoom_flow_est"):
        assert rel(net.act[name].asnumpy(), ref[name]) < 1e-4, name
    flow = out["flow_est_crop"].asnumpy()
    assert np.abs(flow - ref["flow_est"]).max() <= 1e-4 * max(1.0, np.abs(ref["flow_est"]).max())
    mism = float(np.mean(out["mask_observed_pred"].asnumpy() != ref["mask_observed_pred"]))
    assert mism < 1e-4, mism
    assert rel(out["se3"].asnumpy(), ref["se3"]) < 1e-4
    pose = net.pose_update(ctx.array(npd["src_pose"])).asnumpy()
    assert rel(pose, ref["pose_est"]) < 1e-4


def test_config5_rgbd_input_fp16_conv_path(ctx):
    d = synthetic.make_batch(2, seed=505, n_frames=1)
    B = 2
    cfg = default_config()
    cfg.network.INPUT_DEPTH = True
    cfg.network.FP16_CONV = True
    net = deepIM_flownet().get_symbol(cfg)
    assert net.cin == 10
    params = net.init_weights(cfg, seed=55)
    assert params["flow_conv1_weight"].shape == (64, 10, 7, 7)
    net.bind(ctx, B, params)
    assert net.cin_pad == 16
    npd = {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][0], "mask_observed": d["mask_observed"],
           "mask_rendered": d["mask_rendered"][0], "src_pose": d["src_pose"][0],
           "depth_observed": d["depth_gt_observed"], "depth_rendered": d["depth_rendered"][0]}
    pose = net.refine_iteration({k: ctx.array(v) for k, v in npd.items()}).asnumpy()
    emu = opipe.refine_iteration(params, npd, d["K"], MEANS_REV, cfg.dataset.trans_means, cfg.dataset.trans_stds,
                                 cfg.network.ROT_COORD, fp16_conv=True)
    assert emu["net_input"].shape[1] == 10
    # the front end writes conv1's fp16 pixel records directly: their values are the fp32 net input rounded once (RNE), bit for bit
    assert net._input_live_h16
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), opipe.q16(emu["net_input"]))
    np.testing.assert_array_equal(net.act["zoom_factor"].asnumpy(), emu["zoom_factor"])
    assert np.abs(emu["net_input"][:, 6:8]).max() > 0                                   # the depth channels carry data
    c = net.act["conv6_1"].asnumpy()
    assert np.abs(c - emu["conv6_1"]).max() <= 2e-3 * np.abs(emu["conv6_1"]).max()
    assert rel(net.act["se3"].asnumpy(), emu["se3"]) < 1e-3
    assert rel(pose, emu["pose_est"]) < 1e-4


def test_config5_rgbd_input_split_fp16_meets_the_fp32_bar(ctx):
    """Config 5's input as written (INPUT_DEPTH, C_in = 10) with the split-fp16 convs instead of plain fp16: conv1 takes the
    fp32 kernel with the split16 epilogue (the patch kernel is built for 8 channels), and the result meets the fp32 bar
    (se3 / pose ≤ 1e-4 of the fp32 oracle) that plain fp16 cannot."""
    d = synthetic.make_batch(2, seed=506, n_frames=1)
    B = 2
    cfg = default_config()
    cfg.network.INPUT_DEPTH = True
    cfg.network.X3_CONV = True
    net = deepIM_flownet().get_symbol(cfg)
    assert net.cin == 10 and net.x3_conv
    params = net.init_weights(cfg, seed=56)
    net.bind(ctx, B, params)
    assert "flow_conv1" not in net.packed_x3
    npd = {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][0], "mask_observed": d["mask_observed"],
           "mask_rendered": d["mask_rendered"][0], "src_pose": d["src_pose"][0],
           "depth_observed": d["depth_gt_observed"], "depth_rendered": d["depth_rendered"][0]}
    pose = net.refine_iteration({k: ctx.array(v) for k, v in npd.items()}).asnumpy()
    ref = opipe.refine_iteration(params, npd, d["K"], MEANS_REV, cfg.dataset.trans_means, cfg.dataset.trans_stds, cfg.network.ROT_COORD)
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), ref["net_input"])
    assert rel(net.act["conv6_1"].asnumpy(), ref["conv6_1"]) < 1e-5
    assert rel(net.act["se3"].asnumpy(), ref["se3"]) < 1e-4
    assert rel(pose, ref["pose_est"]) < 1e-4
    st = __import__("ctypes").c_int(0)
    lib.deepim_zoom_status(ctx.handle, __import__("ctypes").byref(st))
    assert st.value == 0            # nothing saturated: depth values (metres) times 16 stay far inside fp16's range
