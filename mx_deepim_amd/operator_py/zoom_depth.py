"""ZoomDepth CustomOp — mirror of deepim/operator_py/zoom_depth.py (Prop :52-78, Operator :17-49):
resample the two depth maps with a given zoom factor.  Compute: deepim_zoom_depth_forward (HIP)."""
from .. import mx
from ..runtime import lib
from ._common import targets


class ZoomDepthOperator(mx.operator.CustomOp):
    def __init__(self, height, width):
        super(ZoomDepthOperator, self).__init__()
        self.height = height
        self.width = width

    def forward(self, is_train, req, in_data, out_data, aux):
        ctx = in_data[0].context
        batch_size = in_data[0].shape[0]
        t = targets(out_data, req)
        lib.deepim_zoom_depth_forward(ctx.handle, in_data[0], in_data[1], in_data[2], t[0], t[1], batch_size,
                                      self.height, self.width)
        self.assign(out_data[0], req[0], t[0])
        self.assign(out_data[1], req[1], t[1])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(3):
            self.assign(in_grad[i], req[i], 0)


@mx.operator.register("ZoomDepth")
class ZoomDepthProp(mx.operator.CustomOpProp):
    def __init__(self, width=640, height=480):
        super(ZoomDepthProp, self).__init__(True)
        self.height = int(height)
        self.width = int(width)

    def list_arguments(self):
        return ["zoom_factor", "depth_observed", "depth_rendered"]

    def list_outputs(self):
        return ["zoom_depth_observed", "zoom_depth_rendered"]

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[1], in_shape[2]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 3, [dtype] * 2, []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomDepthOperator(self.height, self.width)
