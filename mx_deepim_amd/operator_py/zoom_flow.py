"""ZoomFlow CustomOp — mirror of deepim/operator_py/zoom_flow.py (Prop :79-112, Operator :21-77):
forward zoom of GT flow (÷wx) + binarised weights, or inverse zoom of predicted flow (×wx).
Compute: deepim_zoom_flow_forward (HIP)."""
from .. import mx
from ..runtime import lib
from ._common import istrue, targets


class ZoomFlowOperator(mx.operator.CustomOp):
    def __init__(self, height, width, b_inv_zoom):
        super(ZoomFlowOperator, self).__init__()
        self.height = height
        self.width = width
        self.b_inv_zoom = b_inv_zoom

    def forward(self, is_train, req, in_data, out_data, aux):
        ctx = in_data[0].context
        batch_size = in_data[0].shape[0]
        zf = in_data[0].asnumpy()
        assert (zf[:, 0] == zf[:, 1]).all(), "wx and wy should be equal"
        t = targets(out_data, req)
        if self.b_inv_zoom:
            lib.deepim_zoom_flow_forward(ctx.handle, in_data[0], in_data[1], None, t[0], None, 1, batch_size,
                                         self.height, self.width)
        else:
            lib.deepim_zoom_flow_forward(ctx.handle, in_data[0], in_data[1], in_data[2], t[0], t[1], 0, batch_size,
                                         self.height, self.width)
        for i in range(len(out_data)):
            self.assign(out_data[i], req[i], t[i])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(len(in_grad)):
            self.assign(in_grad[i], req[i], 0)


@mx.operator.register("ZoomFlow")
class ZoomFlowProp(mx.operator.CustomOpProp):
    def __init__(self, width=640, height=480, b_inv_zoom="False"):
        super(ZoomFlowProp, self).__init__(True)
        self.height = int(height)
        self.width = int(width)
        self.b_inv_zoom = istrue(b_inv_zoom)

    def list_arguments(self):
        return ["zoom_factor", "flow"] if self.b_inv_zoom else ["zoom_factor", "flow", "flow_weights"]

    def list_outputs(self):
        return ["zoom_flow"] if self.b_inv_zoom else ["zoom_flow", "zoom_flow_weights"]

    def infer_shape(self, in_shape):
        return in_shape, list(in_shape[1:]), []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * len(in_type), [dtype] * (len(in_type) - 1), []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomFlowOperator(self.height, self.width, self.b_inv_zoom)
