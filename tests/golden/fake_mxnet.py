"""A numpy-backed stand-in for the slice of `mxnet` (and `cv2`) that the reference's
deepim/operator_py/zoom_*.py files touch, so those files can be imported UNMODIFIED from
/root/reference and their own arithmetic lines executed to make golden vectors.

TEST INFRASTRUCTURE ONLY: used by tests/golden/make_zoom_golden.py and tests/golden/make_ops_golden.py
(transform3d.py, flow_updater.py, group_picker.py) in the build container (the reference checkout does
not exist on the GPU box). Nothing under mx_deepim_amd/ imports it.

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
# accum: how the fake's reductions (batch_dot, sum) add up — MXNet hands them to a BLAS / mshadow reduce whose order is
# not specified, so fixtures are made under two readings: "seq" = unfused float32, left to right; "f64" = float64
# accumulation rounded once.  py2_shapes: NDArray.shape items behave like Python-2-era integers (group_picker.py).
_STATE = {"promotion": "legacy", "sample": True, "affines": [], "accum": "seq", "py2_shapes": False}


def set_accum(mode):
    assert mode in ("seq", "f64")
    _STATE["accum"] = mode


def set_py2_shapes(on):
    _STATE["py2_shapes"] = bool(on)


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
_BIN_UFUNCS = {np.add: lambda a, b: a + b, np.subtract: lambda a, b: a - b, np.multiply: lambda a, b: a * b,
               np.true_divide: lambda a, b: a / b, np.power: lambda a, b: a ** b}
_CMP_UFUNCS = {np.equal: lambda a, b: a == b, np.not_equal: lambda a, b: a != b, np.less: lambda a, b: a < b,
               np.less_equal: lambda a, b: a <= b, np.greater: lambda a, b: a > b, np.greater_equal: lambda a, b: a >= b}
_UNARY_UFUNCS = (np.sqrt, np.exp, np.log, np.absolute, np.negative, np.sin, np.cos)


class LegacyF32(object):
    """np.float32 scalar with NumPy-1.x promotion against Python / int64 / float64 scalars."""
    __slots__ = ("v",)

    def __init__(self, v):
        self.v = f32(v)

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        """np.sqrt / np.exp of a float32 scalar stay float32 in every NumPy; a binary ufunc reached from a NumPy
        scalar's operator (np.int64(3) - x) goes through the same promotion table as the Python operators."""
        if method != "__call__" or kw:
            return NotImplemented
        if len(inputs) == 1 and ufunc in _UNARY_UFUNCS:
            return LegacyF32(ufunc(self.v))
        if len(inputs) == 2 and (ufunc in _BIN_UFUNCS or ufunc in _CMP_UFUNCS):
            a, b = inputs
            swap = a is not self
            o = a if swap else b
            if ufunc in _CMP_UFUNCS:
                ov, _ = self._other(o)
                x, y = (f64(ov), f64(self.v)) if swap else (f64(self.v), f64(ov))
                return bool(_CMP_UFUNCS[ufunc](x, y))
            return self._bin(o, _BIN_UFUNCS[ufunc], swap)
        return NotImplemented

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
        if isinstance(o, np.ndarray) and o.ndim > 0:
            # NumPy-1.x value-based casting: an array of float kind keeps its dtype against any float scalar
            if o.dtype not in (np.dtype(f32), np.dtype(f64)):
                raise TypeError("LegacyF32: array operand of dtype %s not modelled" % o.dtype)
            arr = np.asarray(o).view(np.ndarray)
            r = fn(arr, self.v.astype(arr.dtype)) if swap else fn(self.v.astype(arr.dtype), arr)
            return _as_legacy(r.astype(arr.dtype))
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
    def __abs__(self): return LegacyF32(abs(self.v))
    def __pow__(self, o): return self._bin(o, lambda a, b: a ** b)
    def __rpow__(self, o): return self._bin(o, lambda a, b: a ** b, True)

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
    def __int__(self): return int(self.v)
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

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        """NumPy-1.x value-based casting for array ⊕ scalar: a NumPy float64 / int64 *scalar* never widens a float32
        array (NumPy 2 would: np.float64 scalars are strong there), so such scalars are handed on as Python
        scalars, which NEP 50 treats as weak — the same result dtype as the old rule for these operands."""
        conv = []
        for x in inputs:
            if isinstance(x, LegacyF32):
                conv.append(x.v)
            elif isinstance(x, LegacyArray):
                conv.append(np.asarray(x).view(np.ndarray))
            elif isinstance(x, np.float64):
                conv.append(float(x))
            elif isinstance(x, np.integer) and not isinstance(x, np.bool_):
                conv.append(int(x))
            else:
                conv.append(x)
        if "out" in kw:
            kw["out"] = tuple(np.asarray(o).view(np.ndarray) if isinstance(o, LegacyArray) else o for o in kw["out"])
        r = getattr(ufunc, method)(*conv, **kw)
        if isinstance(r, np.ndarray) and r.ndim > 0 and _STATE["promotion"] == "legacy":
            return r.view(LegacyArray)
        if isinstance(r, np.ndarray) and r.ndim == 0:
            r = r[()]
        if isinstance(r, np.float32) and _STATE["promotion"] == "legacy":
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


# ------------------------------------------------------------------------ Python-2-era ints ----
class Py2Int(object):
    """A shape entry as group_picker.py:22-56 needs it.  That file computes `output_shape[1] /= self.group_num` on
    np.copy(shape) and slices with `input_shape[1] / self.group_num` under `from __future__ import division`: on the
    NumPy of its day (< 1.10: in-place true-divide into an int array cast back unsafely; < 1.12: float indices
    accepted with a DeprecationWarning) both yield the integer quotient; on NumPy 2 / Python 3 both raise.  This
    object is an integer whose true division, when exact, is again that integer — and refuses inexact division."""
    __slots__ = ("i",)

    def __init__(self, i):
        self.i = int(i)

    def __index__(self): return self.i
    def __int__(self): return self.i
    def __repr__(self): return "Py2Int(%d)" % self.i
    def __hash__(self): return hash(self.i)
    def _v(self, o): return o.i if isinstance(o, Py2Int) else o

    def __truediv__(self, o):
        o = int(self._v(o))
        if self.i % o:
            raise TypeError("Py2Int: inexact division %d / %d (the reference asserts divisibility)" % (self.i, o))
        return Py2Int(self.i // o)

    def __mod__(self, o): return self.i % int(self._v(o))
    def __mul__(self, o): return Py2Int(self.i * int(self._v(o)))
    __rmul__ = __mul__
    def __add__(self, o): return Py2Int(self.i + int(self._v(o)))
    __radd__ = __add__
    def __sub__(self, o): return Py2Int(self.i - int(self._v(o)))
    def __eq__(self, o): return self.i == self._v(o)
    def __ne__(self, o): return self.i != self._v(o)
    def __lt__(self, o): return self.i < self._v(o)
    def __le__(self, o): return self.i <= self._v(o)
    def __gt__(self, o): return self.i > self._v(o)
    def __ge__(self, o): return self.i >= self._v(o)


class Py2Shape(tuple):
    def __new__(cls, shape):
        return tuple.__new__(cls, [Py2Int(v) for v in shape])


def _int_shape(shape):
    if isinstance(shape, np.ndarray):
        shape = shape.tolist()
    return tuple(int(v) for v in shape)


# ------------------------------------------------------------------------------ NDArray ----
class NDArray(object):
    """float32 device array stand-in (MXNet's default dtype)."""

    def __init__(self, a, ctx=None):
        self.a = np.ascontiguousarray(a, dtype=f32)
        self.context = ctx

    @property
    def shape(self):
        if _STATE["py2_shapes"]:
            return Py2Shape(self.a.shape)
        return self.a.shape

    @property
    def dtype(self): return self.a.dtype
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

    def __radd__(self, o): return NDArray(self._scalar(o) + self.a, self.context)
    def __rsub__(self, o): return NDArray(self._scalar(o) - self.a, self.context)
    def __rmul__(self, o): return NDArray(self._scalar(o) * self.a, self.context)
    def __rtruediv__(self, o): return NDArray(self._scalar(o) / self.a, self.context)
    def __neg__(self): return NDArray(-self.a, self.context)

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
    return NDArray(np.zeros(_int_shape(shape) if not isinstance(shape, int) else (shape,), f32), ctx)


def _nd_ones(shape, ctx=None, dtype=f32):
    return NDArray(np.ones(_int_shape(shape), f32), ctx)


def _nd_zeros_like(x, ctx=None, dtype=f32):
    return NDArray(np.zeros_like(x.a), x.context)


def _slice_axis(x, axis, begin, end):
    idx = [slice(None)] * x.a.ndim
    idx[axis] = slice(begin, end)
    return NDArray(x.a[tuple(idx)].copy(), x.context)


def _expand_dims(x, axis):
    return NDArray(np.expand_dims(x.a, axis), x.context)


def _nd_add(a, b):
    return NDArray(a.a + b.a, a.context)          # elementwise float32 with NumPy broadcasting (broadcast_add)


def _broadcast_mul(a, b):
    return NDArray(a.a * b.a, a.context)


def _seq_matmul(A, B):
    """float32 GEMM, the K terms of every output added one at a time, left to right, no fusion"""
    out = np.zeros(A.shape[:-1] + (B.shape[-1],), f32)
    for k in range(A.shape[-1]):
        out = (out + (A[..., :, k:k + 1] * B[..., k:k + 1, :]).astype(f32)).astype(f32)
    return out


def _batch_dot(a, b, transpose_a=False, transpose_b=False):
    """mx.nd.batch_dot: per-sample GEMM (MXNet: a batched BLAS sgemm; accumulation order unspecified → two readings)."""
    A = np.swapaxes(a.a, 1, 2) if transpose_a else a.a
    B = np.swapaxes(b.a, 1, 2) if transpose_b else b.a
    if _STATE["accum"] == "f64":
        return NDArray(np.matmul(A.astype(f64), B.astype(f64)).astype(f32), a.context)
    return NDArray(_seq_matmul(A, B), a.context)


def _nd_sum(x, axis=None):
    if _STATE["accum"] == "f64":
        return NDArray(x.a.astype(f64).sum(axis=axis).astype(f32), x.context)
    acc = np.add.accumulate(x.a, axis=axis, dtype=f32)      # strictly sequential float32
    return NDArray(np.take(acc, -1, axis=axis), x.context)


def _nd_transpose(x, axes=None):
    return NDArray(np.transpose(x.a, axes), x.context)


def _nd_exp(x):
    return NDArray(np.exp(x.a), x.context)        # MXNet: expf per element; NumPy's float32 exp (third-party either way)


def _nd_concat(*xs, **kw):
    return NDArray(np.concatenate([x.a for x in xs], axis=kw.get("dim", 1)), xs[0].context)


def _nd_split(x, axis=1, num_outputs=1):
    return [NDArray(p, x.context) for p in np.split(x.a, num_outputs, axis=axis)]


def _nd_tile(x, reps):
    return NDArray(np.tile(x.a, reps), x.context)


def _scalar_or_nd(v):
    return v.a if isinstance(v, NDArray) else f32(float(v))      # *_scalar ops: the attr is cast to DType


def _minimum(a, b):
    return NDArray(np.minimum(_scalar_or_nd(a), _scalar_or_nd(b)), a.context if isinstance(a, NDArray) else b.context)


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
    return NDArray(np.maximum(_scalar_or_nd(a), _scalar_or_nd(b)), a.context if isinstance(a, NDArray) else b.context)


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
                     ("BilinearSampler", _bilinear_sampler), ("round", _round), ("maximum", _maximum),
                     ("minimum", _minimum), ("ones", _nd_ones), ("zeros_like", _nd_zeros_like), ("slice_axis", _slice_axis),
                     ("expand_dims", _expand_dims), ("add", _nd_add), ("broadcast_mul", _broadcast_mul),
                     ("batch_dot", _batch_dot), ("sum", _nd_sum), ("transpose", _nd_transpose), ("exp", _nd_exp),
                     ("concat", _nd_concat), ("split", _nd_split), ("tile", _nd_tile)):
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
