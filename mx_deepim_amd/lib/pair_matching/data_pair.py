"""Loader-side label generation on the device — mirror of the label half of
lib/pair_matching/data_pair.py:133-248 (`get_data_pair_train_batch`) and of the helpers it calls in
lib/utils/image.py (`get_pair_mask` :229-390, `get_pair_flow` :402-450) and data_pair.py
(`get_point_cloud_observed`).  File decoding / resizing / augmentation (cv2) stay with the caller: this takes the decoded
per-pair tensors already resident in HBM and produces every label tensor the training graph consumes, with no host
round trip:

    rot, trans              calc_RT_delta(pose_rendered, pose_observed, ROT_COORD, ROT_TYPE)     deepim_calc_rt_delta_ex
    flow, flow_weights      calc_flow → (B,2,H,W), weights by TRAIN.FLOW_WEIGHT_TYPE             deepim_pair_flow_labels
    mask_rendered           depth_rendered with values > 0.2 set to 1                            deepim_depth_clip_mask
    mask_observed           TRAIN.INIT_MASK: mask_gt | box_gt | box_rendered                     deepim_mask_box_forward
    point_cloud_observed    R·X + T of the sampled model points                                  deepim_points_transform
"""
import ctypes

import numpy as np

from ...runtime import lib
from .RT_transform import calc_RT_delta_batch

FLOW_WEIGHT_TYPE_CODE = {"all": 0, "viz": 1, "valid": 2}


def get_point_cloud_observed(config, points_model, pose_observed):
    """data_pair.py `get_point_cloud_observed`, batched: points_model (B,3,N), pose_observed (B,3,4) device → (B,3,N)."""
    ctx = pose_observed.context
    B, _, N = points_model.shape
    out = ctx.empty((B, 3, N))
    lib.deepim_points_transform(ctx.handle, out, points_model, pose_observed, B, N)
    return out


def get_pair_flow(batch, config):
    """image.py:402-450 on resident depths/poses → (flow (B,2,H,W), flow_weights (B,2,H,W))."""
    wtype = config.TRAIN.FLOW_WEIGHT_TYPE
    if wtype not in FLOW_WEIGHT_TYPE_CODE:
        raise Exception("Unknown FLOW_WEIGHT_TYPE: {}".format(wtype))
    dr = batch["depth_rendered"]
    ctx = dr.context
    B, _, H, W = dr.shape
    flow, wts = ctx.empty((B, 2, H, W)), ctx.empty((B, 2, H, W))
    K = np.ascontiguousarray(config.dataset.INTRINSIC_MATRIX, np.float32).reshape(3, 3)
    lib.deepim_pair_flow_labels(ctx.handle, flow, wts, dr, batch["depth_gt_observed"], batch["pose_rendered"],
                                batch["pose_observed"], K, ctypes.c_float(3e-3),
                                1 if config.network.get("STANDARD_FLOW_REP", False) else 0, FLOW_WEIGHT_TYPE_CODE[wtype], B, H, W)
    return flow, wts


def get_pair_mask(batch, config):
    """image.py:229-390, train phase, on resident tensors → (mask_observed, mask_gt_observed, mask_rendered)."""
    dr, gt = batch["depth_rendered"], batch["mask_gt_observed"]
    ctx = dr.context
    B, _, H, W = dr.shape
    n = B * H * W
    mask_rendered = ctx.empty((B, 1, H, W))
    lib.deepim_depth_clip_mask(ctx.handle, mask_rendered, dr, ctypes.c_float(0.2), n)
    init = config.TRAIN.INIT_MASK
    if init == "mask_gt":
        mask_observed = gt.copy()
    elif init == "box_gt":
        mask_observed = ctx.empty((B, 1, H, W))
        lib.deepim_mask_box_forward(ctx.handle, mask_observed, gt, B, H, W)
    elif init == "box_rendered":
        # the rectangle of depth_rendered > 0.2 (image.py:270-285; that branch assigns `cur_mask_observed` but appends
        # `mask_observed` — the rectangle is what the code means to hand over)
        fg = ctx.empty((B, 1, H, W))
        lib.deepim_depth_to_mask(ctx.handle, fg, dr, ctypes.c_float(0.2), n)
        mask_observed = ctx.empty((B, 1, H, W))
        lib.deepim_mask_box_forward(ctx.handle, mask_observed, fg, B, H, W)
    else:
        raise Exception("Unknown mask type: {}".format(init))
    if config.TRAIN.get("MASK_DILATE", False):
        raise NotImplementedError("TRAIN.MASK_DILATE: the random cv2 dilation (image.py:287-288) is loader-side "
                                  "augmentation and is not part of the device path")
    return mask_observed, gt, mask_rendered


def get_data_pair_train_batch(batch, config):
    """Device composition of data_pair.py:133-248 given decoded tensors (DeviceArrays):
        image_observed, image_rendered (B,3,H,W); depth_gt_observed, depth_rendered (B,1,H,W) in metres;
        pose_rendered, pose_observed (B,3,4); mask_gt_observed (B,1,H,W) [INPUT_MASK / PRED_MASK];
        depth_observed (B,1,H,W) [INPUT_DEPTH]; point_cloud_model, point_cloud_weights (B,3,N) [SE3_PM_LOSS];
        class_index (host).
    Returns {"data": {...}, "label": {...}} with the reference's keys."""
    n, c = config.network, config
    data = {"image_observed": batch["image_observed"], "image_rendered": batch["image_rendered"],
            "depth_gt_observed": batch["depth_gt_observed"], "class_index": batch.get("class_index"),
            "src_pose": batch["pose_rendered"], "tgt_pose": batch["pose_observed"]}
    if n.INPUT_DEPTH:
        data["depth_observed"], data["depth_rendered"] = batch["depth_observed"], batch["depth_rendered"]
    rot, trans = calc_RT_delta_batch(batch["pose_rendered"], batch["pose_observed"], c.dataset.trans_means,
                                     c.dataset.trans_stds, n.ROT_COORD, n.ROT_TYPE)
    label = {"rot": rot, "trans": trans}
    if n.INPUT_MASK or n.PRED_MASK:
        mask_observed, mask_gt_observed, mask_rendered = get_pair_mask(batch, config)
        label["mask_gt_observed"] = mask_gt_observed
        if n.INPUT_MASK:
            data["mask_observed"], data["mask_rendered"] = mask_observed, mask_rendered
    if n.PRED_FLOW:
        label["flow"], label["flow_weights"] = get_pair_flow(batch, config)
    if c.train_iter.SE3_PM_LOSS:
        label["point_cloud_model"] = batch["point_cloud_model"]
        label["point_cloud_weights"] = batch["point_cloud_weights"]
        label["point_cloud_observed"] = get_point_cloud_observed(config, batch["point_cloud_model"], batch["pose_observed"])
    return {"data": data, "label": label}
