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


def conv2d(x, w, b, stride, pad, slope=1.0, pair_order=False):
    """pair_order: False/0 = (ci, ky, kx); True/1 = (ci/2, ky, kx, ci%2); 2 = (ci/8, ky, kx, s, h), channel 8(ci/8)+s+4h
    — the accumulation orders of the three MI355X conv kernels, see net.c."""
    x, w, b = _c(x), _c(w), _c(b)
    B, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    assert not pair_order or Cin % (8 if int(pair_order) == 2 else 2) == 0
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = np.empty((B, Cout, Ho, Wo), f32)
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
