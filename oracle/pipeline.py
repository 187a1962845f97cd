"""End-to-end CPU oracle of one pose-refinement iteration.  TEST INFRASTRUCTURE ONLY.

Composition follows deepim/symbols/deepIM_flownet.py:548-735 (test graph), :32-169 (get_convs) and
deepim/core/tester.py:340-398 (predict → RT_transform).
"""
import numpy as np

from . import net, se3, zoom

f32 = np.float32
SLOPE = 0.1
ENCODER = [("flow_conv1", 2, 3), ("conv2", 2, 2), ("conv3", 2, 2), ("conv3_1", 1, 1), ("conv4", 2, 1), ("conv4_1", 1, 1),
           ("conv5", 2, 1), ("conv5_1", 1, 1), ("conv6", 2, 1), ("conv6_1", 1, 1)]


def encoder(params, x, nc8=False, conv1_nc8=False):
    """nc8: accumulate every layer in the order the channel-blocked MI355X configuration uses (conv1: channel pairs on
    the NCHW net input, order 1; the rest: order 2 of net.c) instead of the canonical (ci,ky,kx). conv1_nc8: conv1 in order
    2 as well (the opt-in configuration where the zoom front end writes channel-blocked records)."""
    acts = {}
    for li, (name, s, p) in enumerate(ENCODER):
        order = (2 if (li > 0 or conv1_nc8) else 1) if nc8 else 0
        x = net.conv2d(x, params[name + "_weight"], params[name + "_bias"], s, p, SLOPE, pair_order=order)
        acts[name] = x
    return acts


def q16(x):
    """Round to fp16 (RNE) and back — what the fp16 path stores between layers."""
    return np.asarray(x, f32).astype(np.float16).astype(f32)


def encoder_fp16(params, x):
    """Emulation of the fp16 conv path (BASELINE config 5): fp16-rounded activations and weights, fp32
    accumulation, bias + LeakyReLU in fp32, fp16-rounded outputs."""
    acts = {}
    x = q16(x)
    for name, s, p in ENCODER:
        x = q16(net.conv2d(x, q16(params[name + "_weight"]), params[name + "_bias"], s, p, SLOPE))
        acts[name] = x
    return acts


def pose_head(params, feat, zoom_factor):
    fc6 = net.fc(feat.reshape(feat.shape[0], -1), params["fc6_weight"], params["fc6_bias"], SLOPE)
    fc7 = net.fc(fc6, params["fc7_weight"], params["fc7_bias"], SLOPE)
    rot = net.fc(fc7, params["rot_weight"], params["rot_bias"], 1.0)
    tr = net.fc(fc7, params["trans_weight"], params["trans_bias"], 1.0)
    tr = zoom.zoom_trans(zoom_factor, tr, b_inv_zoom=True)
    return fc6, fc7, np.concatenate([rot, tr], axis=1).astype(f32)


def decoder(params, acts):
    P = params
    flow6 = net.conv2d(acts["conv6_1"], P["Convolution1_weight"], P["Convolution1_bias"], 1, 1, 1.0)
    d5 = net.deconv4x4s2_crop(acts["conv6_1"], P["deconv5_weight"], P["deconv5_bias"], 15, 20, (1, 1), SLOPE)
    up6 = net.deconv4x4s2_crop(flow6, P["upsample_flow6to5_weight"], P["upsample_flow6to5_bias"], 15, 20, (1, 1), 1.0)
    concat2 = np.concatenate([acts["conv5_1"], d5, up6], axis=1)
    flow5 = net.conv2d(concat2, P["Convolution2_weight"], P["Convolution2_bias"], 1, 1, 1.0)
    d4 = net.deconv4x4s2_crop(concat2, P["deconv4_weight"], P["deconv4_bias"], 30, 40, (1, 1), SLOPE)
    up5 = net.deconv4x4s2_crop(flow5, P["upsample_flow5to4_weight"], P["upsample_flow5to4_bias"], 30, 40, (1, 1), 1.0)
    concat3 = np.concatenate([acts["conv4_1"], d4, up5], axis=1)
    return {"flow6": flow6, "Concat2": concat2, "flow5": flow5, "Concat3": concat3}


def sigmoid(x):
    x = np.asarray(x, f32)
    return (f32(1.0) / (f32(1.0) + np.exp(-x, dtype=f32))).astype(f32)


def mask_head(params, concat3, zoom_factor, H, W):
    low = net.conv2d(concat3, params["mask_conv3_weight"], params["mask_conv3_bias"], 1, 1, 1.0)
    logits = net.upsample16_crop(low, params["mask_upsampling_weight"], H, W, (8, 8), 1.0)
    prob = sigmoid(logits)
    inv = zoom.zoom_mask_with_factor(zoom_factor, prob, b_inv_zoom=True)
    return low, logits, zoom.roundf(inv)


def flow_head(params, concat3, zoom_factor, H, W, normalize_flow):
    low = net.conv2d(concat3, params["Convolution3_weight"], params["Convolution3_bias"], 1, 1, 1.0)
    zflow = net.upsample16_crop(low, params["upsampling_weight"], H, W, (8, 8), normalize_flow)
    (flow,) = zoom.zoom_flow(zoom_factor, zflow, None, b_inv_zoom=True)
    return low, zflow, flow


def refine_iteration(params, data, K, pixel_means_rev, T_means, T_stds, rot_coord="CAMERA", heads=False,
                     normalize_flow=20.0, fp16_conv=False, nc8=False, conv1_nc8=False):
    """-> dict with net_input, zoom_factor, encoder activations, se3 (B,7), pose_est (B,3,4 float64)."""
    x, zf = zoom.net_input(data["image_observed"], data["image_rendered"], data["mask_observed"], data["mask_rendered"],
                           data["src_pose"], K, pixel_means_rev, data.get("depth_observed"), data.get("depth_rendered"))
    out = {"net_input": x, "zoom_factor": zf}
    acts = encoder_fp16(params, x) if fp16_conv else encoder(params, x, nc8=nc8, conv1_nc8=conv1_nc8)
    out.update(acts)
    if heads:
        dec = decoder(params, acts)
        out.update(dec)
        H, W = x.shape[2:]
        out["mask_lowres"], out["mask_logits"], out["mask_observed_pred"] = mask_head(params, dec["Concat3"], zf, H, W)
        out["flow_lowres"], out["zoom_flow_est"], out["flow_est"] = flow_head(params, dec["Concat3"], zf, H, W, normalize_flow)
    out["fc6"], out["fc7"], out["se3"] = pose_head(params, acts["conv6_1"], zf)
    B = x.shape[0]
    pose = np.zeros((B, 3, 4))
    for b in range(B):
        pose[b] = se3.RT_transform(np.asarray(data["src_pose"][b], f32), out["se3"][b, :4], out["se3"][b, 4:], T_means,
                                   T_stds, rot_coord)
    out["pose_est"] = pose
    return out


def train_iteration(params, data, label, K, pixel_means_rev, T_means, T_stds, rot_coord="CAMERA", lw_pm=0.1,
                    num_3d_sample=3000, normalize_3d=0.1, loss_type="L1", sigma=1.0, pred_flow=False, pred_mask=False,
                    lw_flow=0.25, lw_mask=0.03, normalize_flow=20.0, se3_pm_loss=True, se3_dist_loss=False, lw_rot=0.0, lw_trans=0.0,
                    trans_loss_type="L2", trans_sigma=3.0):
    """Forward + backward of the training graph (deepIM_flownet.py:367-546, losses :170-365; backward = module.backward,
    deepim/core/module.py:1131-1137): the point-matching pose branch, plus — pred_flow / pred_mask — the FlowNetS refinement
    decoder with the flow loss (:183-207) and the mask loss (:314-361). Returns (loss_sum, grads keyed like params, forward
    dict); loss_sum is the point-matching sum (the metric train.py logs). se3_dist_loss adds the rotation distance loss
    1 - (q_gt . q_est)^2 (grad_scale lw_rot) and the translation loss on the ZOOMED deltas (:238-262; label["rot"], label["trans"])."""
    from . import heads
    x, zf = zoom.net_input(data["image_observed"], data["image_rendered"], data["mask_observed"], data["mask_rendered"],
                           data["src_pose"], K, pixel_means_rev, data.get("depth_observed"), data.get("depth_rendered"),
                           mask_gt_observed=label["mask_gt_observed"])
    acts = encoder(params, x)
    B, _, H, W = x.shape
    feat = acts["conv6_1"].reshape(B, -1)
    fc6 = net.fc(feat, params["fc6_weight"], params["fc6_bias"], SLOPE)
    fc7 = net.fc(fc6, params["fc7_weight"], params["fc7_bias"], SLOPE)
    rot = net.fc(fc7, params["rot_weight"], params["rot_bias"], 1.0)
    ztr = net.fc(fc7, params["trans_weight"], params["trans_bias"], 1.0)
    rot_norm = heads.l2_normalize(rot)
    trans_est = zoom.zoom_trans(zf, ztr, b_inv_zoom=True)
    pts = se3.transform3d_forward(label["point_cloud_model"], rot_norm, trans_est, data["src_pose"], T_means, T_stds, rot_coord)
    loss, loss_sum, d_pts = heads.point_matching_loss(pts, label["point_cloud_observed"], label["point_cloud_weights"],
                                                      normalize_3d, loss_type, sigma, lw_pm / num_3d_sample)
    fwd = dict(acts, net_input=x, zoom_factor=zf, fc6=fc6, fc7=fc7, rot=rot, zoom_trans=ztr, rot_norm=rot_norm,
               trans_est=trans_est, points_est=pts, pm_loss=loss)
    g = {}
    d_skip = {}
    d_dec61 = None
    if pred_flow or pred_mask:
        P = params
        dec = decoder(P, acts)
        fwd.update(dec)
        C3, C2 = dec["Concat3"], dec["Concat2"]
        dC3 = np.zeros_like(C3)
        if pred_flow:
            low = net.conv2d(C3, P["Convolution3_weight"], P["Convolution3_bias"], 1, 1, 1.0)
            est = net.upsample16_crop(low, P["upsampling_weight"], H, W, (8, 8), 1.0)                      # flow_est_crop
            zflow, zfw = zoom.zoom_flow(zf, label["flow"], label["flow_weights"], b_inv_zoom=False)
            fl, fl_sum, d_est = heads.flow_loss(est, zflow, zfw, normalize_flow, lw_flow / (480 * 640))
            d_low = net.upsample16_crop_backward(d_est, P["upsampling_weight"], 30, 40, (8, 8), 1.0)
            dx, g["Convolution3_weight"], g["Convolution3_bias"] = net.conv2d_backward(C3, P["Convolution3_weight"], d_low, 1, 1)
            dC3 += dx
            g["upsampling_weight"] = np.zeros_like(P["upsampling_weight"])                                  # lr_mult 0
            fwd.update(flow_lowres=low, flow_est_crop=est, zoom_flow_gt=zflow, zoom_flow_weights=zfw, flow_loss=fl,
                       flow_loss_sum=fl_sum)
        if pred_mask:
            low = net.conv2d(C3, P["mask_conv3_weight"], P["mask_conv3_bias"], 1, 1, 1.0)
            logits = net.upsample16_crop(low, P["mask_upsampling_weight"], H, W, (8, 8), 1.0)
            zgt = zoom.zoom_mask(data["mask_observed"], label["mask_gt_observed"], data["mask_rendered"], data["src_pose"], K)[1]
            # LogisticRegressionOutput backward: grad_scale / num_output * (p - y), num_output = H*W per sample
            prob, d_logits = heads.mask_logistic(logits, zgt, lw_mask / (H * W))
            d_low = net.upsample16_crop_backward(d_logits, P["mask_upsampling_weight"], 30, 40, (8, 8), 1.0)
            dx, g["mask_conv3_weight"], g["mask_conv3_bias"] = net.conv2d_backward(C3, P["mask_conv3_weight"], d_low, 1, 1)
            dC3 += dx
            g["mask_upsampling_weight"] = np.zeros_like(P["mask_upsampling_weight"])
            fwd.update(mask_lowres=low, mask_logits=logits, mask_prob=prob, zoom_mask_gt_observed=zgt)
        # Concat3 = [conv4_1 | lrelu(deconv4) | upsample_flow5to4]
        d_skip["conv4_1"] = np.ascontiguousarray(dC3[:, :512])
        d_d4 = net.lrelu_backward(dC3[:, 512:768], C3[:, 512:768], SLOPE)
        dC2, g["deconv4_weight"], g["deconv4_bias"] = net.deconv4x4s2_crop_backward(C2, P["deconv4_weight"], d_d4)
        d_f5, g["upsample_flow5to4_weight"], g["upsample_flow5to4_bias"] = net.deconv4x4s2_crop_backward(
            dec["flow5"], P["upsample_flow5to4_weight"], dC3[:, 768:770])
        dx, g["Convolution2_weight"], g["Convolution2_bias"] = net.conv2d_backward(C2, P["Convolution2_weight"], d_f5, 1, 1)
        dC2 = (dC2 + dx).astype(f32)
        # Concat2 = [conv5_1 | lrelu(deconv5) | upsample_flow6to5]
        d_skip["conv5_1"] = np.ascontiguousarray(dC2[:, :512])
        d_d5 = net.lrelu_backward(dC2[:, 512:1024], C2[:, 512:1024], SLOPE)
        d_dec61, g["deconv5_weight"], g["deconv5_bias"] = net.deconv4x4s2_crop_backward(acts["conv6_1"], P["deconv5_weight"], d_d5)
        d_f6, g["upsample_flow6to5_weight"], g["upsample_flow6to5_bias"] = net.deconv4x4s2_crop_backward(
            dec["flow6"], P["upsample_flow6to5_weight"], dC2[:, 1024:1026])
        dx, g["Convolution1_weight"], g["Convolution1_bias"] = net.conv2d_backward(acts["conv6_1"], P["Convolution1_weight"], d_f6, 1, 1)
        d_dec61 = (d_dec61 + dx).astype(f32)
        fwd.update(d_Concat3=dC3, d_Concat2=dC2)
    # ---- pose branch
    if se3_pm_loss:
        d_rot_norm, d_trans_est = se3.transform3d_backward(d_pts, label["point_cloud_model"], rot_norm, trans_est, data["src_pose"],
                                                            T_means, T_stds, rot_coord)
        d_ztr = zoom.zoom_trans_backward(zf, d_trans_est, b_inv_zoom=True, b_zoom_grad=False)
    else:
        d_rot_norm, d_ztr = np.zeros_like(rot_norm), np.zeros_like(ztr)
    if se3_dist_loss:
        zoom_trans_gt = zoom.zoom_trans(zf, label["trans"], b_inv_zoom=False)                    # :455-457
        rot_loss, d_q = heads.rot_dist_loss(label["rot"], rot_norm, lw_rot)                      # :240-248
        tl, tl_sum, d_zt = heads.point_matching_loss(ztr.reshape(B, 3, 1), zoom_trans_gt.reshape(B, 3, 1), None, 1.0, trans_loss_type,
                                                     trans_sigma, lw_trans)                       # :250-262
        d_rot_norm = (d_rot_norm + d_q).astype(f32)
        d_ztr = (d_ztr + d_zt.reshape(B, 3)).astype(f32)
        fwd.update(zoom_trans_gt=zoom_trans_gt, rot_loss=rot_loss, trans_loss=tl, trans_loss_sum=tl_sum)
    d_rot = heads.l2_normalize_backward(d_rot_norm, rot)
    dx_r, g["rot_weight"], g["rot_bias"] = net.fc_backward(fc7, params["rot_weight"], d_rot)
    dx_t, g["trans_weight"], g["trans_bias"] = net.fc_backward(fc7, params["trans_weight"], d_ztr)
    d = net.lrelu_backward((dx_r + dx_t).astype(f32), fc7, SLOPE)
    d, g["fc7_weight"], g["fc7_bias"] = net.fc_backward(fc6, params["fc7_weight"], d)
    d = net.lrelu_backward(d, fc6, SLOPE)
    d, g["fc6_weight"], g["fc6_bias"] = net.fc_backward(feat, params["fc6_weight"], d)
    d = d.reshape(acts["conv6_1"].shape)
    if d_dec61 is not None:
        d = (d + d_dec61).astype(f32)
    for li in range(len(ENCODER) - 1, -1, -1):
        name, s, p = ENCODER[li]
        if name in d_skip:
            d = (d + d_skip[name]).astype(f32)
        src = x if li == 0 else acts[ENCODER[li - 1][0]]
        dz = net.lrelu_backward(d, acts[name], SLOPE)
        d, g[name + "_weight"], g[name + "_bias"] = net.conv2d_backward(src, params[name + "_weight"], dz, s, p, need_dx=li > 0)
    return loss_sum, g, fwd


def train_pose_iteration(*args, **kw):
    """The pose branch alone (PRED_FLOW = PRED_MASK = False)."""
    return train_iteration(*args, **kw)
