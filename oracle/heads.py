"""H-group oracle: losses and GroupPicker.  TEST INFRASTRUCTURE ONLY.
Follows deepim/symbols/deepIM_flownet.py:200-207 (flow loss), :265-312 (point matching),
:342-349 (LogisticRegressionOutput) with MXNet's documented abs / square / smooth_l1 / MakeLoss
definitions (third-party: PARITY UNPINNED), and deepim/operator_py/group_picker.py:22-56."""
import numpy as np

f32 = np.float32


def _loss_fn(x, loss_type, sigma):
    if loss_type == "L1":
        return np.abs(x), np.sign(x)
    if loss_type == "L2":
        return x * x, 2 * x
    s2 = f32(sigma) * f32(sigma)
    small = np.abs(x) < f32(1.0) / s2
    f = np.where(small, f32(0.5) * (f32(sigma) * x) * (f32(sigma) * x), np.abs(x) - f32(0.5) / s2)
    df = np.where(small, s2 * x, np.where(x > 0, f32(1), f32(-1)))
    return f.astype(f32), df.astype(f32)


def point_matching_loss(est, gt, weights, normalize, loss_type="L1", sigma=1.0, grad_scale=1.0):
    est, gt = np.asarray(est, f32), np.asarray(gt, f32)
    x = ((est - gt) / f32(normalize)).astype(f32)
    f, df = _loss_fn(x, loss_type, sigma)
    w = np.ones_like(x) if weights is None else np.asarray(weights, f32)
    loss = (w * f).astype(f32)
    d_est = (f32(grad_scale) * w * df / f32(normalize)).astype(f32)
    return loss, float(loss.astype(np.float64).sum()), d_est


def flow_loss(est, gt, weights, normalize_flow, grad_scale=1.0):
    est, gt = np.asarray(est, f32), np.asarray(gt, f32)
    d = (est - gt / f32(normalize_flow)).astype(f32)
    w = np.ones_like(d) if weights is None else np.asarray(weights, f32)
    loss = (w * (d * d)).astype(f32)
    return loss, float(loss.astype(np.float64).sum()), (f32(grad_scale) * w * f32(2) * d).astype(f32)


def mask_logistic(logits, label, grad_scale=1.0):
    logits = np.asarray(logits, f32)
    p = (f32(1.0) / (f32(1.0) + np.exp(-logits, dtype=f32))).astype(f32)
    g = None if label is None else ((p - np.asarray(label, f32)) * f32(grad_scale)).astype(f32)
    return p, g


def group_picker(x, group_idx, group_num):
    x = np.asarray(x, f32)
    cg = x.shape[1] // group_num
    out = np.zeros((x.shape[0], cg) + x.shape[2:], f32)       # group_picker.py:29-31: only the channel axis shrinks
    for b in range(x.shape[0]):
        g = int(np.squeeze(group_idx[b]))
        assert 0 <= g < group_num
        out[b] = x[b, cg * g:cg * (g + 1)]
    return out


def group_picker_backward(out_grad, group_idx, group_num, C):
    og = np.asarray(out_grad, f32)
    cg = C // group_num
    g_in = np.zeros((og.shape[0], C) + og.shape[2:], f32)
    for b in range(og.shape[0]):
        g = int(np.squeeze(group_idx[b]))
        g_in[b, cg * g:cg * (g + 1)] = og[b]
    return g_in


def l2_normalize(x, eps=1e-10):
    """MXNet L2Normalization (instance mode): x / sqrt(sum(x^2) + eps) (deepIM_flownet.py:217)."""
    x = np.asarray(x, f32)
    nrm = np.sqrt(np.sum(x.astype(np.float64) ** 2, axis=1, keepdims=True) + eps)
    return (x / nrm).astype(f32)


def l2_normalize_backward(d_out, x, eps=1e-10):
    x, g = np.asarray(x, np.float64), np.asarray(d_out, np.float64)
    nrm = np.sqrt(np.sum(x ** 2, axis=1, keepdims=True) + eps)
    y = x / nrm
    return ((g - y * np.sum(g * y, axis=1, keepdims=True)) / nrm).astype(f32)


def rot_dist_loss(q_gt, q_est, grad_scale=1.0):
    """deepIM_flownet.py:238-248: 1 - (q_gt . q_est)^2 and its gradient w.r.t. q_est (MakeLoss grad_scale)."""
    q_gt, q_est = np.asarray(q_gt, np.float64), np.asarray(q_est, np.float64)
    dot = np.sum(q_gt * q_est, axis=1)
    return (1 - dot ** 2).astype(f32), (-2 * dot[:, None] * q_gt * grad_scale).astype(f32)
