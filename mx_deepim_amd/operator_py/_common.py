"""Helpers shared by the CustomOp mirrors."""
import ctypes

import numpy as np

from ..runtime import lib

cf = ctypes.c_float


def parse_vec(s, n=None):
    """'[a b c]' → float32 array, the way the reference Props do (np.fromstring(s[1:-1], sep=' '))."""
    if isinstance(s, np.ndarray):
        v = s.astype(np.float32).reshape(-1)
    else:
        v = np.array([float(x) for x in str(s).strip()[1:-1].replace(",", " ").split()], dtype=np.float32)
    if n is not None:
        assert v.size == n, (s, n)
    return np.ascontiguousarray(v)


def istrue(s):
    """How every reference Prop but Transform3D reads a boolean kwarg: `s.lower() == "true"` (zoom_trans.py:81-82, zoom_flow.py:85,
    zoom_mask_with_factor.py:77, flow_updater.py:118) — MXNet hands kwargs over as strings; "1" / "yes" / "on" are False there."""
    return str(s).lower() == "true"


def strtobool(s):
    """distutils.util.strtobool, the one Transform3D uses (transform3d.py:22, :290): y / yes / t / true / on / 1 and their opposites,
    ValueError otherwise."""
    if isinstance(s, bool):
        return s
    s = str(s).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return True
    if s in ("n", "no", "f", "false", "off", "0"):
        return False
    raise ValueError("invalid truth value %r" % (s,))


def targets(out_data, req):
    """Where the kernel should write each output: straight into out_data[i] for write/inplace, a scratch
    array otherwise (then CustomOp.assign adds / drops it)."""
    return [o if r in ("write", "inplace") else o.context.empty(o.shape, o.dtype) for o, r in zip(out_data, req)]


def check_zoom_status(ctx, what):
    st = ctypes.c_int(0)
    lib.deepim_zoom_status(ctx.handle, ctypes.byref(st))
    if st.value & 8:
        # DI_STATUS_X3_SATURATED: reading the word clears it, so it must be reported here, not dropped
        raise FloatingPointError("%s: a split-fp16 (x3) convolution saturated fp16's range earlier on this context — "
                                 "its results are invalid (network.X3_CONV needs activations < 65504/16)" % what)
    if st.value & 1:
        # the reference dies in np.min(nz_x) with this ValueError (zoom_mask.py:55 / zoom_image.py:46)
        raise ValueError("zero-size array to reduction operation minimum which has no identity (%s: "
                         "observed mask/image has no valid pixel)" % what)
    if st.value & 2:
        raise AssertionError("%s: group index out of range" % what)
    if st.value & 4:
        # data_pair.py:98 / utils/image.py:366: np.min of the empty nonzero() of the rendered mask
        raise ValueError("zero-size array to reduction operation minimum which has no identity (%s: "
                         "rendered mask is empty, no box_rendered rectangle)" % what)
