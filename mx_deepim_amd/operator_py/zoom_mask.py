"""ZoomMask CustomOp — mirror of deepim/operator_py/zoom_mask.py (Prop :121-150, Operator :21-118).
zoom mask: bbox of observed-gt and rendered masks + projected object centre → zoom factor; the three
masks are resampled (bilinear, rounded) with it.  Compute: deepim_zoom_mask_forward (HIP)."""
import numpy as np

from .. import mx
from ..runtime import lib
from ._common import check_zoom_status, parse_vec, targets


class ZoomMaskOperator(mx.operator.CustomOp):
    def __init__(self, K, height, width):
        super(ZoomMaskOperator, self).__init__()
        self.K = np.ascontiguousarray(K, dtype=np.float32).reshape(3, 3)
        self.height = height
        self.width = width

    def forward(self, is_train, req, in_data, out_data, aux):
        ctx = in_data[0].context
        batch_size = in_data[0].shape[0]
        t = targets(out_data, req)
        lib.deepim_zoom_mask_forward(ctx.handle, in_data[0], in_data[1], in_data[2], in_data[3], self.K, t[0], t[1],
                                     t[2], t[3], batch_size, self.height, self.width)
        check_zoom_status(ctx, "ZoomMask")
        for i in range(4):
            self.assign(out_data[i], req[i], t[i])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(4):
            self.assign(in_grad[i], req[i], 0)


@mx.operator.register("ZoomMask")
class ZoomMaskProp(mx.operator.CustomOpProp):
    def __init__(self, K, width=640, height=480):
        super(ZoomMaskProp, self).__init__(True)
        self.K = parse_vec(K, 9).reshape([3, 3])
        self.height = int(height)
        self.width = int(width)

    def list_arguments(self):
        return ["mask_observed", "mask_gt_observed", "mask_rendered", "src_pose"]

    def list_outputs(self):
        return ["zoom_mask_observed", "zoom_mask_gt_observed", "zoom_mask_rendered", "zoom_factor"]

    def infer_shape(self, in_shape):
        batch_size = in_shape[0][0]
        out_shape = list(in_shape[:-1])
        out_shape.append([batch_size, 4])
        return in_shape, out_shape, []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 4, [dtype] * 4, []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomMaskOperator(self.K, self.height, self.width)
