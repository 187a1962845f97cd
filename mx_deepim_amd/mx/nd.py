"""`mx.nd` surface: array / zeros / Custom (imperative invocation of a registered CustomOp, the
equivalent of `mx.sym.Custom(..., op_type=...)` bound and run once)."""
import numpy as np

from ..runtime import Context, DeviceArray

NDArray = DeviceArray


def array(source, ctx=None, dtype=np.float32):
    ctx = ctx or Context.default()
    if isinstance(source, DeviceArray):
        return source.copy()
    return ctx.array(np.asarray(source), dtype=dtype)


def zeros(shape, ctx=None, dtype=np.float32):
    return (ctx or Context.default()).zeros(tuple(shape) if not isinstance(shape, int) else (shape,), dtype)


def load(fname):
    """`mx.nd.load`: NDArray-list file -> dict (or list) of host arrays; see lib/utils/ndarray_file.py."""
    from ..lib.utils import ndarray_file
    return ndarray_file.load(fname)


def save(fname, data):
    from ..lib.utils import ndarray_file
    ndarray_file.save(fname, data)


def _attr_to_str(v):
    """MXNet passes every Custom attr as a string; numpy arrays print as '[a b c]' and the Props parse
    them with np.fromstring(s[1:-1], sep=' ') (zoom_mask.py:125)."""
    if isinstance(v, np.ndarray):
        return "[" + " ".join(repr(float(x)) for x in v.reshape(-1)) + "]"
    return str(v)


def Custom(*args, **kwargs):
    from . import operator as op
    op_type = kwargs.pop("op_type")
    kwargs.pop("name", None)
    is_train = bool(kwargs.pop("is_train", False))
    prop_cls = op.get_registered(op_type)
    arg_names_probe = None
    # split tensor kwargs (named inputs) from attrs
    tensors = {k: v for k, v in kwargs.items() if isinstance(v, DeviceArray)}
    attrs = {k: _attr_to_str(v) for k, v in kwargs.items() if not isinstance(v, DeviceArray)}
    prop = prop_cls(**attrs)
    arg_names = prop.list_arguments()
    in_data = list(args)
    for name in arg_names[len(in_data):]:
        if name not in tensors:
            raise TypeError("Custom(%s): missing input %r" % (op_type, name))
        in_data.append(tensors[name])
    if len(in_data) != len(arg_names):
        raise TypeError("Custom(%s): expected inputs %s" % (op_type, arg_names))
    del arg_names_probe
    in_shapes = [list(a.shape) for a in in_data]
    _, out_shapes, aux_shapes = prop.infer_shape(in_shapes)
    in_types = [a.dtype for a in in_data]
    _, out_types, _ = prop.infer_type(in_types)
    ctx = in_data[0].context
    out_data = [ctx.empty(tuple(s), np.dtype(t)) for s, t in zip(out_shapes, out_types)]
    operator = prop.create_operator(ctx, in_shapes, in_types)
    operator.forward(is_train, ["write"] * len(out_data), in_data, out_data, [])
    Custom.last_operator = operator  # lets callers run backward on the same instance
    return out_data[0] if len(out_data) == 1 else out_data
