"""Between-iteration batch update on the device — mirror of the pose/flow half of
lib/pair_matching/batch_updater_py_multi.py (`batchUpdaterPyMulti.forward` :91-328): apply the predicted
delta (RT_transform :179), re-render, recompute the ground-truth delta labels (calc_RT_delta :239), the
K·T matrices (:255-259), the ground-truth flow via lib/flow_c (:276-290) and the rendered mask
(`depth > 0.2`, :261-265), and rewrite the batch.

Everything except the re-render runs as HIP kernels on resident tensors (the reference does it per sample in
numpy with host round trips). The OpenGL renderer is outside the hot path: pass a
`render_machine` whose `render_batch(class_index, poses) -> (image_rendered, depth_rendered)` returns device
arrays already laid out like the network inputs (RGB order, mean-subtracted, NCHW), or pre-stage the frames
in the batch under "next_image_rendered" / "next_depth_rendered".
"""
import ctypes

import numpy as np

from ...config import ROT_COORD_CODE
from ...runtime import lib


class batchUpdaterPyMulti(object):
    def __init__(self, big_cfg, height, width, render_machine=None):
        self.big_cfg = big_cfg
        self.rot_coord = big_cfg.network.ROT_COORD
        self.K = np.ascontiguousarray(big_cfg.dataset.INTRINSIC_MATRIX, dtype=np.float32).reshape(3, 3)
        self.T_means = np.ascontiguousarray(big_cfg.dataset.trans_means, dtype=np.float32)
        self.T_stds = np.ascontiguousarray(big_cfg.dataset.trans_stds, dtype=np.float32)
        self.height = height
        self.width = width
        self.render_machine = render_machine
        self.Kinv = np.ascontiguousarray(np.linalg.inv(self.K))   # float32, as np.linalg.inv(np.matrix(K)) gives
        self._rc = ROT_COORD_CODE[self.rot_coord.lower()]

    def workspace(self, ctx, B):
        """Preallocated outputs for `forward(..., out=ws)`: a resident training step (deepIM_flownet.train_step) updates its batch
        TRAIN_ITER_SIZE - 1 times per step and must not allocate inside the loop. The buffers are rewritten by every call — the
        previous iteration's forward/backward has consumed them by then (module.py:1131-1137 runs the updater after update())."""
        H, W = self.height, self.width
        ws = {"se3": ctx.empty((B, 7)), "src_pose": ctx.empty((B, 3, 4)), "image_rendered": ctx.empty((B, 3, H, W)),
              "depth_rendered": ctx.empty((B, 1, H, W)), "rot": ctx.empty((B, 4)), "trans": ctx.empty((B, 3))}
        if self.big_cfg.network.PRED_FLOW:
            ws.update(KT=ctx.empty((B, 3, 4)), flow=ctx.empty((B, 2, H, W)), valid=ctx.empty((B, 1, H, W)),
                      flow_weights=ctx.empty((B, 2, H, W)))
        if self.big_cfg.network.INPUT_MASK:
            ws["mask_rendered"] = ctx.empty((B, 1, H, W))
        return ws

    def forward(self, data_batch, preds, big_cfg=None, out=None):
        """data_batch: dict name -> DeviceArray with src_pose, tgt_pose (B,3,4) [, depth_gt_observed (B,1,H,W),
        class_index]; preds: dict with rot_est (B,4), trans_est (B,3) (or se3 (B,7)). Returns the updated batch.
        `out`: a `workspace()` whose buffers receive the results instead of fresh allocations."""
        cfg = big_cfg or self.big_cfg
        src_pose, tgt_pose = data_batch["src_pose"], data_batch["tgt_pose"]
        ctx, h = src_pose.context, src_pose.context.handle
        B, H, W = src_pose.shape[0], self.height, self.width
        out = out or {}

        def buf(name, shape):
            return out[name] if name in out else ctx.empty(shape)
        if "se3" in preds:
            se3 = preds["se3"]
        else:
            se3 = buf("se3", (B, 7))
            lib.deepim_copy_channels(h, se3, 7, 0, preds["rot_est"], 4, B, 1)
            lib.deepim_copy_channels(h, se3, 7, 4, preds["trans_est"], 3, B, 1)
        # 1. refined pose (in place when src_pose already is the workspace buffer: the kernel reads a pose before it writes it)
        refined_pose = buf("src_pose", (B, 3, 4))
        lib.deepim_rt_transform(h, refined_pose, None, src_pose, se3, self.T_means, self.T_stds, self._rc, B)
        # 2. re-render at the refined pose (outside the path)
        fused_mask = None
        if self.render_machine is not None and "image_rendered" in out:
            image_rendered, depth_rendered = out["image_rendered"], out["depth_rendered"]
            fused_mask = out.get("mask_rendered") if cfg.network.INPUT_MASK else None
            self.render_machine.render_batch(data_batch.get("class_index"), refined_pose, out=(image_rendered, depth_rendered),
                                             mask_rendered=fused_mask, mask_thresh=0.2,
                                             light_intensity=data_batch.get("light_intensity"))
        elif self.render_machine is not None:
            image_rendered, depth_rendered = self.render_machine.render_batch(data_batch.get("class_index"), refined_pose,
                                                                              light_intensity=data_batch.get("light_intensity"))
        else:
            image_rendered, depth_rendered = data_batch["next_image_rendered"], data_batch["next_depth_rendered"]
        # 3. residual delta = new labels
        rot, trans = buf("rot", (B, 4)), buf("trans", (B, 3))
        lib.deepim_calc_rt_delta(h, rot, trans, refined_pose, tgt_pose, self.T_means, self.T_stds, self._rc, B)
        update_package = {"image_rendered": image_rendered, "depth_rendered": depth_rendered, "src_pose": refined_pose,
                          "rot": rot, "trans": trans}
        # 4./5. K·T and ground-truth flow rendered → observed
        if cfg.network.PRED_FLOW:
            KT = buf("KT", (B, 3, 4))
            lib.deepim_calc_KT(h, KT, refined_pose, tgt_pose, self.K, B)
            flow, valid = buf("flow", (B, 2, H, W)), buf("valid", (B, 1, H, W))
            lib.deepim_flow_forward(h, flow, valid, depth_rendered, data_batch["depth_gt_observed"], KT, self.Kinv, B, H, W)
            flow_weights = buf("flow_weights", (B, 2, H, W))          # np.tile(valid, [1, 2, 1, 1])
            lib.deepim_copy_channels(h, flow_weights, 2, 0, valid, 1, B, H * W)
            lib.deepim_copy_channels(h, flow_weights, 2, 1, valid, 1, B, H * W)
            update_package["flow"] = flow
            update_package["flow_weights"] = flow_weights
        # 6. rendered mask
        if cfg.network.INPUT_MASK:
            if fused_mask is not None:          # written by the render pass itself (depth > 0.2 in its resolve kernel)
                mask = fused_mask
            else:
                mask = buf("mask_rendered", (B, 1, H, W))
                lib.deepim_depth_to_mask(h, mask, depth_rendered, ctypes.c_float(0.2), B * H * W)
            update_package["mask_rendered"] = mask
        return self.update_data_batch(data_batch, update_package)

    def update_data_batch(self, data_batch, update_package):
        new_batch = dict(data_batch)
        for name, value in update_package.items():
            new_batch[name] = value
        return new_batch


def update_test_batch(cfg, data, render_machine, refined_pose, class_index=None, out=None, light_intensity=None):
    """Test-loop update between refinement iterations — mirror of deepim/core/tester.py:420-455 +
    lib/pair_matching/data_pair.py:62-132 (`update_data_batch`): re-render at the refined pose, mask_rendered =
    depth > 0.2, and (TEST.UPDATE_MASK == "box_rendered") mask_observed = rectangle of the new rendered mask. All on the
    device; `out` may carry preallocated image_rendered / depth_rendered / mask_rendered / mask_observed arrays.
    With the lit ModelNet render machine (tester.py:114-172) `light_intensity` = device (B,3) per-sample light colour (the
    reference draws U(0.9, 1.1) per render; None = 1.0)."""
    ctx = refined_pose.context
    B, H, W = refined_pose.shape[0], render_machine.height, render_machine.width
    out = out or {}
    image = out.get("image_rendered") or ctx.empty((B, 3, H, W))
    depth = out.get("depth_rendered") or ctx.empty((B, 1, H, W))
    new = dict(data)
    mask = box = None
    if cfg.network.INPUT_MASK:
        mask = out.get("mask_rendered") or ctx.empty((B, 1, H, W))
        if cfg.network.PRED_MASK:
            if cfg.TEST.UPDATE_MASK == "box_rendered":
                # tester.py:445-449 hands the old mask_rendered over, but update_data_batch (data_pair.py:94-105) ignores
                # it for this mode and draws the rectangle of the NEW rendered mask
                box = out.get("mask_observed") or ctx.empty((B, 1, H, W))
            elif cfg.TEST.UPDATE_MASK != "init":
                raise Exception("Unknown UPDATE_MASK type: {}".format(cfg.TEST.UPDATE_MASK))
    # one fused pass per run of equal class ids: draw, mask_rendered = depth > 0.2, rectangle
    ids = np.zeros(B, np.int64) if class_index is None else np.broadcast_to(
        np.asarray(class_index).astype(np.int64).reshape(-1), (B,)) if np.size(class_index) == 1 else \
        np.asarray(class_index).astype(np.int64).reshape(B)
    b0 = 0
    while b0 < B:
        b1 = b0 + 1
        while b1 < B and ids[b1] == ids[b0]:
            b1 += 1
        render_machine.render_into(image[b0:b1], depth[b0:b1], ids[b0], refined_pose[b0:b1],
                                   mask_rendered=None if mask is None else mask[b0:b1],
                                   mask_box=None if box is None else box[b0:b1], mask_thresh=0.2,
                                   light_intensity=None if light_intensity is None else light_intensity[b0:b1])
        b0 = b1
    new["image_rendered"], new["src_pose"] = image, refined_pose
    if cfg.network.INPUT_DEPTH:
        new["depth_rendered"] = depth
    if mask is not None:
        new["mask_rendered"] = mask
    if box is not None:
        new["mask_observed"] = box
    return new
