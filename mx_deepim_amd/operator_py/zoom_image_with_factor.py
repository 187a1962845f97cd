"""ZoomImageWithFactor CustomOp — mirror of deepim/operator_py/zoom_image_with_factor.py
(Prop :73-104, Operator :20-70): resample observed + rendered RGB with a given zoom factor; the means
are added before and removed after sampling so out-of-frame pixels come out black.
Compute: deepim_zoom_image_with_factor_forward (HIP)."""
import numpy as np

from .. import mx
from ..runtime import lib
from ._common import parse_vec, targets


class ZoomImageWithFactorOperator(mx.operator.CustomOp):
    def __init__(self, height, width, pixel_means, high_light_center):
        super(ZoomImageWithFactorOperator, self).__init__()
        self.height = height
        self.width = width
        self.pixel_means = np.ascontiguousarray(np.asarray(pixel_means, np.float32).reshape(3))
        self.high_light_center = high_light_center

    def forward(self, is_train, req, in_data, out_data, aux):
        ctx = in_data[0].context
        batch_size = in_data[0].shape[0]
        t = targets(out_data, req)
        lib.deepim_zoom_image_with_factor_forward(ctx.handle, in_data[0], in_data[1], in_data[2], self.pixel_means,
                                                  1 if self.high_light_center else 0, t[0], t[1], batch_size,
                                                  self.height, self.width)
        self.assign(out_data[0], req[0], t[0])
        self.assign(out_data[1], req[1], t[1])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(3):
            self.assign(in_grad[i], req[i], 0)


@mx.operator.register("ZoomImageWithFactor")
class ZoomImageWithFactorProp(mx.operator.CustomOpProp):
    def __init__(self, width=640, height=480, pixel_means="[0 0 0]", high_light_center="False"):
        super(ZoomImageWithFactorProp, self).__init__(True)
        self.height = int(height)
        self.width = int(width)
        self.pixel_means = parse_vec(pixel_means, 3)[::-1].copy()
        self.hight_light_center = str(high_light_center).lower() == "true"

    def list_arguments(self):
        return ["zoom_factor", "image_observed", "image_rendered"]

    def list_outputs(self):
        return ["zoom_image_observed", "zoom_image_rendered"]

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[1], in_shape[2]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 3, [dtype] * 2, []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomImageWithFactorOperator(self.height, self.width, self.pixel_means, self.hight_light_center)
