"""ZoomMaskWithFactor CustomOp — mirror of deepim/operator_py/zoom_mask_with_factor.py
(Prop :71-98, Operator :21-68): binarise (>0.2), (inverse-)zoom, round.
Compute: deepim_zoom_mask_with_factor_forward (HIP)."""
from .. import mx
from ..runtime import lib
from ._common import istrue, targets


class ZoomMaskWithFactorOperator(mx.operator.CustomOp):
    def __init__(self, height, width, b_inv_zoom):
        super(ZoomMaskWithFactorOperator, self).__init__()
        self.height = height
        self.width = width
        self.b_inv_zoom = b_inv_zoom

    def forward(self, is_train, req, in_data, out_data, aux):
        ctx = in_data[0].context
        batch_size = in_data[0].shape[0]
        t = targets(out_data, req)
        lib.deepim_zoom_mask_with_factor_forward(ctx.handle, in_data[0], in_data[1], t[0], 1 if self.b_inv_zoom else 0,
                                                 batch_size, self.height, self.width)
        self.assign(out_data[0], req[0], t[0])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        self.assign(in_grad[0], req[0], 0)
        self.assign(in_grad[1], req[1], 0)


@mx.operator.register("ZoomMaskWithFactor")
class ZoomMaskWithFactorProp(mx.operator.CustomOpProp):
    def __init__(self, width=640, height=480, b_inv_zoom="False"):
        super(ZoomMaskWithFactorProp, self).__init__(True)
        self.height = int(height)
        self.width = int(width)
        self.b_inv_zoom = istrue(b_inv_zoom)

    def list_arguments(self):
        return ["zoom_factor", "mask"]

    def list_outputs(self):
        return ["zoom_mask"]

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[1]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype, dtype], [dtype], []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomMaskWithFactorOperator(self.height, self.width, self.b_inv_zoom)
