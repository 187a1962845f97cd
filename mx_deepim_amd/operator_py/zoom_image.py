"""ZoomImage CustomOp — mirror of deepim/operator_py/zoom_image.py (Prop :115-147, Operator :18-113):
no-mask variant, bbox from non-black pixels, computes the zoom factor AND resamples both images.
Compute: deepim_zoom_image_forward (HIP)."""
import numpy as np

from .. import mx
from ..runtime import lib
from ._common import check_zoom_status, parse_vec, targets


class ZoomImageOperator(mx.operator.CustomOp):
    def __init__(self, K, height, width, pixel_means):
        super(ZoomImageOperator, self).__init__()
        self.K = np.ascontiguousarray(K, dtype=np.float32).reshape(3, 3)
        self.height = height
        self.width = width
        self.pixel_means = np.ascontiguousarray(np.asarray(pixel_means, np.float32).reshape(3))

    def forward(self, is_train, req, in_data, out_data, aux):
        ctx = in_data[0].context
        batch_size = in_data[0].shape[0]
        t = targets(out_data, req)
        lib.deepim_zoom_image_forward(ctx.handle, in_data[0], in_data[1], in_data[2], self.K, self.pixel_means, t[0],
                                      t[1], t[2], batch_size, self.height, self.width)
        check_zoom_status(ctx, "ZoomImage")
        for i in range(3):
            self.assign(out_data[i], req[i], t[i])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(3):
            self.assign(in_grad[i], req[i], 0)


@mx.operator.register("ZoomImage")
class ZoomImageProp(mx.operator.CustomOpProp):
    def __init__(self, K, width=640, height=480, pixel_means="[0 0 0]"):
        super(ZoomImageProp, self).__init__(True)
        self.K = parse_vec(K, 9).reshape([3, 3])
        self.height = int(height)
        self.width = int(width)
        # config means are BGR-ordered, the tensors RGB-ordered (zoom_image.py:122-124)
        self.pixel_means = parse_vec(pixel_means, 3)[::-1].copy()

    def list_arguments(self):
        return ["image_observed", "image_rendered", "src_pose"]

    def list_outputs(self):
        return ["zoom_image_observed", "zoom_image_rendered", "zoom_factor"]

    def infer_shape(self, in_shape):
        batch_size = in_shape[0][0]
        out_shape = list(in_shape[:-1])
        out_shape.append([batch_size, 4])
        return in_shape, out_shape, []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 3, [dtype] * 3, []

    def create_operator(self, ctx, shapes, dtypes):
        return ZoomImageOperator(self.K, self.height, self.width, self.pixel_means)
