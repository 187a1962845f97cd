"""GroupPicker CustomOp — mirror of deepim/operator_py/group_picker.py (Prop :59-86, Operator :15-56):
pick the per-class channel group.  Compute: deepim_group_picker_forward / _backward (HIP)."""
from .. import mx
from ..runtime import lib
from ._common import check_zoom_status, targets


def _rows_cols(shape):
    """(B, C, ...) → (B, C·inner): the picked channel group [g·C/G, (g+1)·C/G) of a sample is one contiguous run of
    (C/G)·inner elements, so trailing axes (group_picker.py:29-39 slices axis 1 of any rank) fold into the channel count."""
    cols = 1
    for d in shape[1:]:
        cols *= int(d)
    return int(shape[0]), cols


class GroupPickerOperator(mx.operator.CustomOp):
    def __init__(self, group_num):
        super(GroupPickerOperator, self).__init__()
        self.group_num = group_num

    def forward(self, is_train, req, in_data, out_data, aux):
        ctx = in_data[0].context
        B, C = _rows_cols(in_data[0].shape)
        assert in_data[0].shape[1] % self.group_num == 0       # group_picker.py:33
        t = targets(out_data, req)
        lib.deepim_group_picker_forward(ctx.handle, t[0], in_data[0], in_data[1], self.group_num, B, C)
        check_zoom_status(ctx, "GroupPicker")
        self.assign(out_data[0], req[0], t[0])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        ctx = in_data[0].context
        B, C = _rows_cols(in_data[0].shape)
        g = in_grad[0] if req[0] in ("write", "inplace") else ctx.empty(in_grad[0].shape)
        lib.deepim_group_picker_backward(ctx.handle, g, out_grad[0], in_data[1], self.group_num, B, C)
        check_zoom_status(ctx, "GroupPicker")
        self.assign(in_grad[0], req[0], g)
        self.assign(in_grad[1], req[1], 0)


@mx.operator.register("GroupPicker")
class GroupPickerProp(mx.operator.CustomOpProp):
    def __init__(self, group_num=1):
        super(GroupPickerProp, self).__init__(True)
        self.group_num = int(group_num)

    def list_arguments(self):
        return ["input_data", "group_idx"]

    def list_outputs(self):
        return ["picked_data"]

    def infer_shape(self, in_shape):
        out = list(in_shape[0])
        out[1] = out[1] // self.group_num
        return in_shape, [out], []

    def infer_type(self, in_type):
        dtype = in_type[0]
        return [dtype, dtype], [dtype], []

    def create_operator(self, ctx, shapes, dtypes):
        return GroupPickerOperator(self.group_num)
