from .deepIM_flownet import deepIM_flownet  # noqa: F401
