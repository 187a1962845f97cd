"""A numpy-backed stand-in for the slice of `mxnet` (and `cv2`) that the reference's
deepim/operator_py/zoom_*.py files touch, so those files can be imported UNMODIFIED from
/root/reference and their own arithmetic lines executed to make golden vectors.

TEST INFRASTRUCTURE ONLY: used by tests/golden/make_zoom_golden.py in the build container (the
reference checkout does not exist on the GPU box). Nothing under mx_deepim_amd/ imports it.

Two things live here:

1. NumPy-1.x ("legacy", pre-NEP-50) scalar promotion.  The reference ran under MXNet 1.2 (2018,
   numpy < 1.17); this container has NumPy 2.2, where `np.float32(x) / 640` stays float32.  Under
   legacy promotion a float32 *scalar* combined with a Python int/float (or a NumPy int64/float64
   scalar) gives float64 — NEP 50's own "old behaviour" table.  `asnumpy()` therefore returns a
   `LegacyArray` whose float32 items come out as `LegacyF32`, a scalar that implements exactly that
   rule and refuses (TypeError) anything it does not model.  float64 / int64 scalars behave the same
   in both NumPy generations, so they stay plain NumPy scalars.  `set_promotion("numpy2")` switches the
   wrapper off and the same reference lines then run with this container's NumPy-2 semantics.

2. GridGenerator(affine) / BilinearSampler / round, the third-party MXNet 1.2 operators (not vendored
   in the reference): restated literally from src/operator/grid_generator-inl.h and
   bilinear_sampler.cc with a materialised (B,2,H,W) grid, independently of oracle/zoom.py (which
   uses separable per-axis taps).  They stay "third-party, unpinned"; what the fixtures pin is the
   reference's own Python around them.
"""
import sys
import types

import numpy as np

f32, f64 = np.float32, np.float64
_STATE = {"promotion": "legacy", "sample": True, "affines": []}


def set_promotion(mode):
    assert mode in ("legacy", "numpy2")
    _STATE["promotion"] = mode


def set_sampling(on):
    """off: BilinearSampler returns zeros (factor-only runs over many cases)."""
    _STATE["sample"] = bool(on)


def captured_affines(clear=True):
    out = list(_STATE["affines"])
    if clear:
        _STATE["affines"].clear()
    return out


# ------------------------------------------------------------------ legacy float32 scalar ----
class LegacyF32(object):
    """np.float32 scalar with NumPy-1.x promotion against Python / int64 / float64 scalars."""
    __array_ufunc__ = None          # make np.int64.__sub__(LegacyF32) return NotImplemented
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = f32(v)

    @staticmethod
    def _other(o):
        """-> (value, is_f32)"""
        if isinstance(o, LegacyF32):
            return o.v, True
        if isinstance(o, (bool, np.bool_)):
            raise TypeError("LegacyF32: bool operand not modelled")
        if isinstance(o, (int, float, np.integer, np.float64)):
            return f64(o), False    # python int -> int64 -> promote(float32,int64)=float64; python float -> float64
        if isinstance(o, np.float32):
            return o, True
        raise TypeError("LegacyF32: operand %r not modelled" % type(o))

    def _bin(self, o, fn, swap=False):
        ov, same = self._other(o)
        if same:
            a, b = (ov, self.v) if swap else (self.v, ov)
            return LegacyF32(fn(f32(a), f32(b)))
        a, b = (ov, f64(self.v)) if swap else (f64(self.v), ov)
        return f64(fn(a, b))

    def __add__(self, o): return self._bin(o, lambda a, b: a + b)
    def __radd__(self, o): return self._bin(o, lambda a, b: a + b, True)
    def __sub__(self, o): return self._bin(o, lambda a, b: a - b)
    def __rsub__(self, o): return self._bin(o, lambda a, b: a - b, True)
    def __mul__(self, o): return self._bin(o, lambda a, b: a * b)
    def __rmul__(self, o): return self._bin(o, lambda a, b: a * b, True)
    def __truediv__(self, o): return self._bin(o, lambda a, b: a / b)
    def __rtruediv__(self, o): return self._bin(o, lambda a, b: a / b, True)
    def __neg__(self): return LegacyF32(-self.v)

    def _cmp(self, o, fn):
        ov, _ = self._other(o)
        return bool(fn(f64(self.v), f64(ov)))   # exact either way

    def __eq__(self, o): return self._cmp(o, lambda a, b: a == b)
    def __ne__(self, o): return self._cmp(o, lambda a, b: a != b)
    def __lt__(self, o): return self._cmp(o, lambda a, b: a < b)
    def __le__(self, o): return self._cmp(o, lambda a, b: a <= b)
    def __gt__(self, o): return self._cmp(o, lambda a, b: a > b)
    def __ge__(self, o): return self._cmp(o, lambda a, b: a >= b)
    __hash__ = None

    def __float__(self): return float(self.v)
    def __repr__(self): return "LegacyF32(%r)" % float(self.v)
    def __format__(self, spec): return format(float(self.v), spec)


class LegacyArray(np.ndarray):
    """ndarray whose float32 scalar items are LegacyF32 (array-level promotion with Python scalars is
    value-based in NumPy 1.x and gives the same dtypes as NumPy 2 for the expressions in zoom_*.py)."""
    __array_priority__ = 15.0      # np.dot(K, src_pose[:, 3]) must hand back a LegacyArray too

    def __getitem__(self, idx):
        r = np.ndarray.__getitem__(self, idx)
        if isinstance(r, np.float32):
            return LegacyF32(r)
        return r


def _as_legacy(a):
    a = np.array(a, copy=True)
    if _STATE["promotion"] == "legacy":
        return a.view(LegacyArray)
    return a


def _plain(x):
    if isinstance(x, NDArray):
        return x.a
    if isinstance(x, LegacyF32):
        return x.v
    if isinstance(x, LegacyArray):
        return np.asarray(x).view(np.ndarray)
    return x


# ------------------------------------------------------------------------------ NDArray ----
class NDArray(object):
    """float32 device array stand-in (MXNet's default dtype)."""

    def __init__(self, a, ctx=None):
        self.a = np.ascontiguousarray(a, dtype=f32)
        self.context = ctx

    @property
    def shape(self): return self.a.shape
    def asnumpy(self): return _as_legacy(self.a)
    def reshape(self, shape): return NDArray(self.a.reshape(shape), self.context)
    def copy(self): return NDArray(self.a.copy(), self.context)

    @staticmethod
    def _scalar(o):
        o = _plain(o)
        if isinstance(o, np.ndarray):
            return o.astype(f32)
        return f32(float(o))          # MXNet *_scalar ops: the attr is parsed to double, cast to DType

    def __add__(self, o): return NDArray(self.a + self._scalar(o), self.context)
    def __sub__(self, o): return NDArray(self.a - self._scalar(o), self.context)
    def __mul__(self, o): return NDArray(self.a * self._scalar(o), self.context)
    def __truediv__(self, o): return NDArray(self.a / self._scalar(o), self.context)

    def __iadd__(self, o): self.a += self._scalar(o); return self
    def __isub__(self, o): self.a -= self._scalar(o); return self
    def __imul__(self, o): self.a *= self._scalar(o); return self
    def __itruediv__(self, o): self.a /= self._scalar(o); return self

    def __getitem__(self, idx):
        v = self.a[idx]
        if isinstance(v, np.ndarray):
            return NDArray.__new_view(v, self.context)
        return NDArray(np.array([v], f32), self.context)

    @staticmethod
    def __new_view(v, ctx):
        n = NDArray.__new__(NDArray)
        n.a = v            # a view: in-place ops on a slice write through (zoom_flow.py:62-64)
        n.context = ctx
        return n

    def __setitem__(self, idx, val):
        val = _plain(val)
        if not isinstance(val, np.ndarray):
            val = f32(float(val))     # python float / float64 / LegacyF32 -> rounded once to float32
        self.a[idx] = val


def _nd_array(src, ctx=None, dtype=f32):
    if isinstance(src, NDArray):
        return NDArray(src.a.copy(), ctx)
    if isinstance(src, (list, tuple)):
        src = [[float(v) for v in row] if isinstance(row, (list, tuple)) else float(row) for row in src]
        return NDArray(np.array(src, dtype=f64).astype(f32), ctx)     # python floats -> float32 once
    return NDArray(np.asarray(_plain(src)).astype(f32), ctx)


def _nd_zeros(shape, ctx=None, dtype=f32):
    return NDArray(np.zeros(tuple(shape), f32), ctx)


def _grid_generator(data, transform_type="affine", target_shape=None):
    """grid_generator-inl.h (affine): grid_dst rows x_d = -1 + (i % W)·(2/(W-1)), y_d = -1 + (i / W)·(2/(H-1)), 1;
    out = data(B,2,3) · grid_dst(3,HW) as a float32 GEMM with K = 3, terms accumulated left to right, unfused."""
    assert transform_type == "affine"
    H, W = target_shape
    A = data.a.reshape(-1, 2, 3)
    _STATE["affines"].append(A.copy())
    i = np.arange(H * W, dtype=np.int64)
    xd = (f32(-1.0) + (i % W).astype(f32) * f32(2.0 / (W - 1))).astype(f32)
    yd = (f32(-1.0) + (i // W).astype(f32) * f32(2.0 / (H - 1))).astype(f32)
    out = np.zeros((A.shape[0], 2, H * W), f32)
    for b in range(A.shape[0]):
        for r in range(2):
            acc = (A[b, r, 0] * xd).astype(f32)
            acc = (acc + (A[b, r, 1] * yd).astype(f32)).astype(f32)
            acc = (acc + (A[b, r, 2] * f32(1.0)).astype(f32)).astype(f32)
            out[b, r] = acc
    return NDArray(out.reshape(-1, 2, H, W), data.context)


def _bilinear_sampler(data, grid):
    """bilinear_sampler.cc BilinearSamplerForward (CPU), per output pixel, DType = float."""
    x = data.a
    if not _STATE["sample"]:
        return NDArray(np.zeros_like(x), data.context)
    g = grid.a
    B, C, iH, iW = x.shape
    oH, oW = g.shape[2], g.shape[3]
    out = np.zeros((B, C, oH, oW), f32)
    for n in range(B):
        y_real = ((g[n, 1] + f32(1)) * f32(iH - 1) / f32(2)).astype(f32)
        x_real = ((g[n, 0] + f32(1)) * f32(iW - 1) / f32(2)).astype(f32)
        ok = np.isfinite(y_real) & np.isfinite(x_real)
        yr = np.where(ok, y_real, f32(-8)).clip(-8, iH + 8)
        xr = np.where(ok, x_real, f32(-8)).clip(-8, iW + 8)
        tly = np.floor(yr).astype(np.int64)
        tlx = np.floor(xr).astype(np.int64)
        tly_w = (1.0 - (y_real - tly.astype(f32)).astype(f32).astype(f64)).astype(f32)   # 1.0 is a double literal
        tlx_w = (1.0 - (x_real - tlx.astype(f32)).astype(f32).astype(f64)).astype(f32)

        def between(v, lo, hi):
            return (v >= lo) & (v <= hi)

        def tap(dy, dx):
            yy, xx = tly + dy, tlx + dx
            inside = ok & between(xx, 0, iW - 1) & between(yy, 0, iH - 1)
            v = x[n][:, yy.clip(0, iH - 1), xx.clip(0, iW - 1)]
            return np.where(inside[None], v, f32(0))

        tl, tr, bl, br = tap(0, 0), tap(0, 1), tap(1, 0), tap(1, 1)
        yw, xw = tly_w[None], tlx_w[None]
        # float*float*float ; float*float*(double) ; float*(double)*float ; float*(double)*(double); sum left to right
        t1 = ((tl * yw).astype(f32) * xw).astype(f32)
        t2 = (tr * yw).astype(f32).astype(f64) * (1.0 - xw.astype(f64))
        t3 = (bl.astype(f64) * (1.0 - yw.astype(f64))) * xw.astype(f64)
        t4 = (br.astype(f64) * (1.0 - yw.astype(f64))) * (1.0 - xw.astype(f64))
        out[n] = (((t1.astype(f64) + t2) + t3) + t4).astype(f32)
    return NDArray(out, data.context)


def _round(x):
    """mshadow_op::round = C roundf: half away from zero."""
    a = x.a
    t = np.trunc(a)
    return NDArray(np.where(np.abs(a - t) >= f32(0.5), t + np.sign(a), t).astype(f32), x.context)


def _maximum(a, b):
    return NDArray(np.maximum(a.a, b.a), a.context)


# -------------------------------------------------------------------- operator protocol ----
class CustomOp(object):
    def __init__(self):
        pass

    def assign(self, dst, req, src):
        if req == "null":
            return
        s = _plain(src)
        if req in ("write", "inplace"):
            dst.a[...] = s
        elif req == "add":
            dst.a[...] += s


class CustomOpProp(object):
    def __init__(self, need_top_grad=True):
        self.need_top_grad_ = need_top_grad


REGISTRY = {}


def register(name):
    def deco(cls):
        REGISTRY[name] = cls
        return cls
    return deco


class Context(object):
    def __init__(self, kind, idx=0):
        self.kind, self.idx = kind, idx


def install():
    """Put fake `mxnet` and `cv2` modules into sys.modules (idempotent)."""
    mx = types.ModuleType("mxnet")
    nd = types.ModuleType("mxnet.ndarray")
    for name, fn in (("array", _nd_array), ("zeros", _nd_zeros), ("GridGenerator", _grid_generator),
                     ("BilinearSampler", _bilinear_sampler), ("round", _round), ("maximum", _maximum)):
        setattr(nd, name, fn)
    nd.NDArray = NDArray
    op = types.ModuleType("mxnet.operator")
    op.CustomOp, op.CustomOpProp, op.register = CustomOp, CustomOpProp, register
    mx.nd = mx.ndarray = nd
    mx.operator = op
    mx.cpu = lambda i=0: Context("cpu", i)
    mx.gpu = lambda i=0: Context("gpu", i)
    sys.modules["mxnet"] = mx
    sys.modules["mxnet.ndarray"] = nd
    sys.modules["mxnet.operator"] = op
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    return mx


def run_op(op_type, inputs, out_shapes, **attrs):
    """mx.nd.Custom-style call of a registered reference operator: attrs are passed as strings, exactly as
    MXNet hands them to the Prop; returns the list of output numpy arrays."""
    prop = REGISTRY[op_type](**{k: str(v) for k, v in attrs.items()})
    ctx = Context("cpu")
    opr = prop.create_operator(ctx, None, None)
    in_data = [NDArray(np.asarray(x, f32), ctx) for x in inputs]
    out_data = [NDArray(np.zeros(s, f32), ctx) for s in out_shapes]
    opr.forward(False, ["write"] * len(out_data), in_data, out_data, [])
    return [o.a.copy() for o in out_data], opr
