"""`mx.operator` surface used by deepim/operator_py/*.py: CustomOp / CustomOpProp / register
(protocol described in SURVEY §8b-B1; MXNet python/mxnet/operator.py is third-party)."""
import ctypes

import numpy as np

from ..runtime import DeviceArray, lib

_REGISTRY = {}


class CustomOp(object):
    def forward(self, is_train, req, in_data, out_data, aux):
        raise NotImplementedError

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        raise NotImplementedError

    def assign(self, dst, req, src):
        """req ∈ {'null','write','inplace','add'}; src may be a DeviceArray, numpy array or scalar."""
        if req == "null":
            return
        if req in ("write", "inplace"):
            if isinstance(src, DeviceArray):
                if src.ptr != dst.ptr:
                    dst.copyfrom(src)
            elif np.isscalar(src) and src == 0:
                lib.deepim_memset(dst.context.handle, dst, 0, dst.nbytes)
            else:
                dst.copyfrom(src)
        elif req == "add":
            if not isinstance(src, DeviceArray):
                if np.isscalar(src) and src == 0:
                    return
                src = dst.context.array(np.broadcast_to(np.asarray(src, np.float32), dst.shape))
            lib.deepim_axpy(dst.context.handle, dst, src, ctypes.c_float(1.0), dst.size)
        else:
            raise ValueError("unknown req %r" % (req,))


class CustomOpProp(object):
    def __init__(self, need_top_grad=False):
        self.need_top_grad_ = need_top_grad

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[0]] * len(self.list_outputs()), []

    def infer_type(self, in_type):
        return in_type, [in_type[0]] * len(self.list_outputs()), []

    def list_outputs(self):
        return ["output"]

    def list_arguments(self):
        return ["data"]

    def list_auxiliary_states(self):
        return []

    def declare_backward_dependency(self, out_grad, in_data, out_data):
        deps = []
        if self.need_top_grad_:
            deps.extend(out_grad)
        deps.extend(in_data)
        deps.extend(out_data)
        return deps

    def create_operator(self, ctx, in_shapes, in_dtypes):
        return CustomOp()


def register(reg_name):
    """`@mx.operator.register("ZoomMask")` — op_type string → Prop class."""

    def do_register(prop_cls):
        _REGISTRY[reg_name] = prop_cls
        return prop_cls

    return do_register


def get_registered(reg_name):
    return _REGISTRY[reg_name]


def registered_ops():
    return sorted(_REGISTRY)
