"""Transform3D CustomOp — mirror of deepim/operator_py/transform3d.py (Prop :284-308, Operator :25-281):
quaternion → R_delta, composed with pose_src per ROT_COORD, applied to (B,3,N) model points; analytic
backward to quaternion and translation.  Compute: deepim_transform3d_forward / _backward (HIP)."""
import numpy as np

from .. import mx
from ..config import ROT_COORD_CODE
from ..runtime import lib
from ._common import parse_vec, strtobool, targets


class transform3dOperator(mx.operator.CustomOp):
    def __init__(self, T_means=None, T_stds=None, rot_coord="MODEL", projection_2d=False):
        super(transform3dOperator, self).__init__()
        self.T_means = np.ascontiguousarray(T_means, dtype=np.float32).reshape(3)
        self.T_stds = np.ascontiguousarray(T_stds, dtype=np.float32).reshape(3)
        self._projection_2d = projection_2d
        self.rot_coord = rot_coord
        assert not projection_2d, "NOT_IMPLEMENTED"
        if rot_coord.lower() not in ROT_COORD_CODE:
            raise Exception("Unknown rot_coord in transform3d operator: {}".format(rot_coord))
        self._rc = ROT_COORD_CODE[rot_coord.lower()]

    def _check(self, in_data):
        batch_size = in_data[0].shape[0]
        rotation, T_delta = in_data[1], in_data[2]
        assert rotation.shape[0] == batch_size and T_delta.shape[0] == batch_size, \
            "rotation.shape[0]:{} vs batch_size:{}, translation.shape[0]:{} vs batch_size:{}".format(
                rotation.shape[0], batch_size, T_delta.shape[0], batch_size)
        if rotation.shape[1] == 3:
            raise Exception("NOT_IMPLEMENTED")
        if rotation.shape[1] != 4:
            raise Exception("UNKNOWN ROTATION REPRESENTATION {}".format(rotation.shape[1]))
        n = int(np.prod(in_data[0].shape[1:])) // 3
        return batch_size, n

    def forward(self, is_train, req, in_data, out_data, aux):
        ctx = in_data[0].context
        batch_size, n = self._check(in_data)
        t = targets(out_data, req)
        lib.deepim_transform3d_forward(ctx.handle, t[0], in_data[0], in_data[1], in_data[2], in_data[3], self.T_means,
                                       self.T_stds, self._rc, batch_size, n)
        self.assign(out_data[0], req[0], t[0])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        ctx = in_data[0].context
        batch_size, n = self._check(in_data)
        gq = in_grad[1] if req[1] in ("write", "inplace") else ctx.empty(in_grad[1].shape)
        gt = in_grad[2] if req[2] in ("write", "inplace") else ctx.empty(in_grad[2].shape)
        lib.deepim_transform3d_backward(ctx.handle, gq, gt, out_grad[0], in_data[0], in_data[1], in_data[2], in_data[3],
                                        self.T_means, self.T_stds, self._rc, batch_size, n)
        self.assign(in_grad[0], req[0], 0)
        self.assign(in_grad[1], req[1], gq)
        self.assign(in_grad[2], req[2], gt)
        self.assign(in_grad[3], req[3], 0)


@mx.operator.register("Transform3D")
class transform3DProp(mx.operator.CustomOpProp):
    def __init__(self, T_means, T_stds, rot_coord="MODEL", b_project_2d="False"):
        super(transform3DProp, self).__init__(True)
        self.T_means = parse_vec(T_means, 3)
        self.T_stds = parse_vec(T_stds, 3)
        self.rot_coord = rot_coord
        self._project_2d = strtobool(b_project_2d)

    def list_arguments(self):
        return ["point_cloud", "rotation", "translation", "pose_src"]

    def list_outputs(self):
        return ["transformed_3d_points"]

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[0]], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype] * 4, [dtype], []

    def create_operator(self, ctx, shapes, dtypes):
        return transform3dOperator(self.T_means, self.T_stds, self.rot_coord, self._project_2d)
