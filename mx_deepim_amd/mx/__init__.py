"""Minimal stand-in for the slice of the `mxnet` Python API the DeepIM CustomOps touch
(SURVEY §8b-B1): `mx.operator.{CustomOp, CustomOpProp, register}`, `mx.nd.{array, zeros, Custom}`,
`mx.gpu(i)`.  NDArrays are `DeviceArray`s over hipMalloc'ed memory; there is no `mx.cpu()` compute
context — this package has no CPU path.
"""
from . import nd, operator  # noqa: F401
from ..runtime import Context


def gpu(device_id=0):
    return Context.get(device_id)


def cpu(device_id=0):
    raise RuntimeError("mx_deepim_amd has no CPU compute context; use mx.gpu(i)")
