"""N-group oracle wrappers (ctypes over oracle/net.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

f32 = np.float32
_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_net.so")
_dll = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "net.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])


def _lib():
    global _dll
    if _dll is None:
        if not os.path.exists(_SO):
            build()
        _dll = ctypes.CDLL(_SO)
    return _dll


def _p(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=f32)


# bench.py's cpu_baseline sets this: canonical-order convolutions then run on the cache-blocked OpenMP build
# (oracle_conv2d_blocked: the same fmaf chain per output, bit-identical, laid out for the host cores)
BLOCKED = False


def omp_threads():
    return int(_lib().oracle_omp_threads())


def set_omp_threads(n):
    _lib().oracle_set_omp_threads(int(n))


def conv2d(x, w, b, stride, pad, slope=1.0, pair_order=False):
    """pair_order: False/0 = (ci, ky, kx); True/1 = (ci/2, ky, kx, ci%2); 2 = (ci/8, ky, kx, s, h), channel 8(ci/8)+s+4h
    — the accumulation orders of the three MI355X conv kernels, see net.c."""
    x, w, b = _c(x), _c(w), _c(b)
    B, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    assert not pair_order or Cin % (8 if int(pair_order) == 2 else 2) == 0
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = np.empty((B, Cout, Ho, Wo), f32)
    if BLOCKED and not pair_order:
        _lib().oracle_conv2d_blocked(_p(out), _p(x), _p(w), _p(b), B, Cin, H, W, Cout, kh, kw, stride, pad, ctypes.c_float(slope))
        return out
    _lib().oracle_conv2d_order(_p(out), _p(x), _p(w), _p(b), B, Cin, H, W, Cout, kh, kw, stride, pad, ctypes.c_float(slope),
                               int(pair_order))
    return out


def deconv4x4s2_crop(x, w, b, Ho, Wo, crop=(1, 1), slope=1.0):
    x, w, b = _c(x), _c(w), _c(b)
    B, Cin, H, W = x.shape
    Cout = w.shape[1]
    out = np.empty((B, Cout, Ho, Wo), f32)
    _lib().oracle_deconv4x4s2_crop(_p(out), _p(x), _p(w), _p(b), B, Cin, H, W, Cout, Ho, Wo, crop[0], crop[1],
                                   ctypes.c_float(slope))
    return out


def upsample16_crop(x, w, Ho, Wo, crop=(8, 8), scale=1.0):
    x, w = _c(x), _c(w)
    B, C, H, W = x.shape
    out = np.empty((B, C, Ho, Wo), f32)
    _lib().oracle_upsample16_crop(_p(out), _p(x), _p(w), B, C, H, W, Ho, Wo, crop[0], crop[1], ctypes.c_float(scale))
    return out


def fc(x, w, b, slope=1.0):
    x, w, b = _c(x), _c(w), _c(b)
    B, I = x.shape
    O = w.shape[0]
    out = np.empty((B, O), f32)
    _lib().oracle_fc(_p(out), _p(x), _p(w), _p(b), B, I, O, ctypes.c_float(slope))
    return out


def bilinear_upsample_weights(C, k=32):
    """`_init_bilinear` (deepIM_flownet.py:808-822 → MXNet Initializer._init_bilinear)."""
    w = np.zeros(C * k * k, dtype=f32)
    f = np.ceil(k / 2.0)
    c = (2 * f - 1 - f % 2) / (2.0 * f)
    for i in range(w.size):
        x = i % k
        y = (i // k) % k
        w[i] = (1 - abs(x / f - c)) * (1 - abs(y / f - c))
    return w.reshape(C, 1, k, k)


# ------------------------------------------------------------------------------------------------ backward ----
def lrelu_backward(dy, y, slope):
    """MXNet LeakyReLU(leaky) gradient from the saved OUTPUT: dy where y > 0, slope*dy elsewhere."""
    dy, y = np.asarray(dy, f32), np.asarray(y, f32)
    return np.where(y > 0, dy, dy * f32(slope)).astype(f32)


def conv2d_backward(x, w, dy, stride, pad, need_dx=True):
    """Gradients of Convolution (bias + no activation): -> (dx or None, dw, db)."""
    x, w, dy = _c(x), _c(w), _c(dy)
    B, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    dw, db = np.empty_like(w), np.empty((Cout,), f32)
    _lib().oracle_conv2d_wgrad(_p(dw), _p(db), _p(x), _p(dy), B, Cin, H, W, Cout, kh, kw, stride, pad)
    dx = None
    if need_dx:
        dx = np.empty_like(x)
        _lib().oracle_conv2d_dgrad(_p(dx), _p(dy), _p(w), B, Cin, H, W, Cout, kh, kw, stride, pad)
    return dx, dw, db


def fc_backward(x, w, dy):
    x, w, dy = _c(x), _c(w), _c(dy)
    B, I = x.shape
    O = w.shape[0]
    dx, dw, db = np.empty_like(x), np.empty_like(w), np.empty((O,), f32)
    _lib().oracle_fc_backward(_p(dx), _p(dw), _p(db), _p(dy), _p(x), _p(w), B, I, O)
    return dx, dw, db


def sgd_mom_update(w, mom, g, lr, wd, momentum, rescale=1.0, clip=None):
    """MXNet sgd_mom_update: mom = momentum*mom - lr*(rescale*g [clipped] + wd*w); w += mom (float32)."""
    w, mom, g = np.asarray(w, f32), np.asarray(mom, f32), np.asarray(g, f32)
    gg = (g * f32(rescale)).astype(f32)
    if clip is not None and clip > 0:
        gg = np.clip(gg, -f32(clip), f32(clip))
    mom2 = (f32(momentum) * mom - f32(lr) * (gg + f32(wd) * w).astype(f32)).astype(f32)
    return (w + mom2).astype(f32), mom2


def deconv4x4s2_crop_backward(x, w, dy, crop=(1, 1)):
    """Gradients of Deconvolution k4 s2 + Crop given the gradient of the CROPPED output: -> (dx, dw, db)."""
    x, w, dy = _c(x), _c(w), _c(dy)
    B, Cin, H, W = x.shape
    Cout, Ho, Wo = w.shape[1], dy.shape[2], dy.shape[3]
    dx, dw, db = np.empty_like(x), np.empty_like(w), np.empty((Cout,), f32)
    _lib().oracle_deconv4x4s2_crop_backward(_p(dx), _p(dw), _p(db), _p(x), _p(w), _p(dy), B, Cin, H, W, Cout, Ho, Wo, crop[0],
                                            crop[1])
    return dx, dw, db


def upsample16_crop_backward(dy, w, H, W, crop=(8, 8), scale=1.0):
    dy, w = _c(dy), _c(w)
    B, C, Ho, Wo = dy.shape
    dx = np.empty((B, C, H, W), f32)
    _lib().oracle_upsample16_crop_backward(_p(dx), _p(dy), _p(w), B, C, H, W, Ho, Wo, crop[0], crop[1], ctypes.c_float(scale))
    return dx
