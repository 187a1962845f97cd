"""CustomOp mirrors of deepim/operator_py/*.py — importing the package registers every op_type."""
from . import (flow_updater, group_picker, transform3d, zoom_depth, zoom_flow, zoom_image,  # noqa: F401
               zoom_image_with_factor, zoom_mask, zoom_mask_with_factor, zoom_trans)
