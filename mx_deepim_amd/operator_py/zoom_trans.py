"""ZoomTrans CustomOp — mirror of deepim/operator_py/zoom_trans.py (Prop :77-103, Operator :16-74):
scale (vx, vy) of the translation by wx (inverse) or 1/wx; optional gradient scaling.
Compute: deepim_zoom_trans_forward / _backward (HIP)."""
from .. import mx
from ..runtime import lib
from ._common import istrue, targets


class ZoomTransOperator(mx.operator.CustomOp):
    def __init__(self, b_inv_zoom, b_zoom_grad):
        super(ZoomTransOperator, self).__init__()
        self.b_inv_zoom = b_inv_zoom
        self.b_zoom_grad = b_zoom_grad

    def forward(self, is_train, req, in_data, out_data, aux):
        ctx = in_data[0].context
        batch_size = in_data[0].shape[0]
        t = targets(out_data, req)
        lib.deepim_zoom_trans_forward(ctx.handle, in_data[0], in_data[1], t[0], 1 if self.b_inv_zoom else 0, batch_size)
        self.assign(out_data[0], req[0], t[0])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        ctx = in_data[0].context
        batch_size = in_data[0].shape[0]
        g = ctx.empty(in_grad[1].shape) if req[1] not in ("write", "inplace") else in_grad[1]
        lib.deepim_zoom_trans_backward(ctx.handle, in_data[0], out_grad[0], g, 1 if self.b_inv_zoom else 0,
                                       1 if self.b_zoom_grad else 0, batch_size)
        self.assign(in_grad[0], req[0], 0)
        self.assign(in_grad[1], req[1], g)


@mx.operator.register("ZoomTrans")
class ZoomTransProp(mx.operator.CustomOpProp):
    def __init__(self, b_inv_zoom="False", b_zoom_grad="False"):
        super(ZoomTransProp, self).__init__(True)
        self.b_inv_zoom = istrue(b_inv_zoom)
        self.b_zoom_grad = istrue(b_zoom_grad)

    def list_arguments(self):
        return ["zoom_factor", "trans_delta"]

    def list_outputs(self):
        return ["zoom_trans_delta"]

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[1]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype, dtype], [dtype], []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomTransOperator(self.b_inv_zoom, self.b_zoom_grad)
