"""FlowUpdater CustomOp — mirror of deepim/operator_py/flow_updater.py (Prop :109-151, Operator :20-107):
on-graph ground-truth flow from two depth maps and two poses (integer flow from rounded, clamped
projections).  Registered but not referenced by the symbol in the reference either.
Compute: deepim_flow_updater_forward (HIP)."""
import ctypes

import numpy as np

from .. import mx
from ..runtime import lib
from ._common import istrue, parse_vec, targets


class flowUpdaterOperator(mx.operator.CustomOp):
    def __init__(self, K, Kinv, thresh, batch_size, height, width, wh_rep):
        super(flowUpdaterOperator, self).__init__()
        self.K = np.ascontiguousarray(K, dtype=np.float32).reshape(3, 3)
        self.thresh = thresh
        self.batch_size = batch_size
        self.height = height
        self.width = width
        self.wh_rep = wh_rep

    def forward(self, is_train, req, in_data, out_data, aux):
        ctx = in_data[0].context
        batch_size = in_data[0].shape[0]
        t = targets(out_data, req)
        lib.deepim_flow_updater_forward(ctx.handle, t[0], t[1], in_data[0], in_data[1], in_data[2], in_data[3], self.K,
                                        ctypes.c_float(self.thresh), 1 if self.wh_rep else 0, batch_size, self.height,
                                        self.width)
        self.assign(out_data[0], req[0], t[0])
        self.assign(out_data[1], req[1], t[1])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(4):
            self.assign(in_grad[i], req[i], 0)


@mx.operator.register("FlowUpdater")
class flowUpdaterProp(mx.operator.CustomOpProp):
    def __init__(self, K, thresh=3e-3, batch_size=4, height=480, width=640, wh_rep="False"):
        super(flowUpdaterProp, self).__init__(True)
        self.K = parse_vec(K, 9).reshape([3, 3])
        self.Kinv = np.linalg.inv(self.K.astype(np.float64))
        self.thresh = float(thresh)
        self.batch_size = int(batch_size)
        self.height = int(height)
        self.width = int(width)
        self.wh_rep = istrue(wh_rep)

    def list_arguments(self):
        return ["depth_src", "depth_tgt", "pose_src", "pose_tgt"]

    def list_outputs(self):
        return ["flow", "flow_weights"]

    def infer_shape(self, in_shape):
        b, _, h, w = in_shape[0]
        return in_shape, [[b, 2, h, w], [b, 2, h, w]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 4, [dtype] * 2, []

    def create_operator(self, ctx, shapes, dtypes):
        return flowUpdaterOperator(self.K, self.Kinv, self.thresh, self.batch_size, self.height, self.width, self.wh_rep)
