"""mx_deepim_amd — MI355X (gfx950) implementation of mx-DeepIM's render-and-compare inner loop.

Host code is Python calling hand-written HIP kernels through the C ABI declared in
``include/deepim_hip.h`` (ctypes, no PyTorch / MXNet at run time).  The package mirrors the
reference's operator surface for the hot path only:

    mx_deepim_amd.mx             minimal ``mx.operator`` / ``mx.nd`` shim (CustomOp protocol)
    mx_deepim_amd.operator_py    ZoomMask, ZoomImage, ..., Transform3D (deepim/operator_py/*.py)
    mx_deepim_amd.lib.flow_c     gpu_flow / gpu_flow_wrapper (lib/flow_c/gpu_flow.pyx, flow.py:19-23)
    mx_deepim_amd.lib.pair_matching  device RT_transform, calc_flow (lib/pair_matching/*.py)
    mx_deepim_amd.symbols        deepIM_flownet test graph as a fused device pipeline
"""
from .runtime import Context, DeviceArray, lib, LibraryMissing  # noqa: F401

__all__ = ["Context", "DeviceArray", "lib", "LibraryMissing"]
