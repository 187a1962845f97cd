"""F1 pinned against the reference's OWN kernel: /root/reference/lib/flow_c/gpu_flow_kernel.cu compiled for gfx950 as
test-only infrastructure (oracle/Makefile: hipify-perl + hipcc → oracle/_ref/libref_flow_{nofma,fma}.so, git-ignored,
shipped to the GPU box prebuilt) and run on the same MI355X as `_flow` / deepim_flow_forward.

Two builds of the one reference source:
  * nofma (-ffp-contract=off): every a*b+c rounded twice — the order our kernel and oracle/flow.py:gpu_flow restate.
    Bar: flow and valid BIT-EXACT.
  * fma (-ffp-contract=fast): what nvcc's default --fmad=true does to the same source on the reference's own hardware
    (mul+add contracted; the exact contraction choice is the compiler's).  Bar: valid flags differ on < 1e-4 of the
    pixels (reprojection-threshold ties), flow within 1e-4 relative on the pixels whose flags agree — north_star's
    flow tolerance.  The measured differences are printed.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle import flow as oflow
from mx_deepim_amd.lib.flow_c.flow import gpu_flow
from mx_deepim_amd.runtime import lib

pytestmark = pytest.mark.gpu
REF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
SYM = "_Z5_flowPfS_S_S_S_S_iiii"   # void _flow(float*,float*,float*,float*,float*,float*,int,int,int,int) — gpu_flow.hpp:1-3 (C++ linkage there)


def _ref_flow(which, depth_src, depth_tgt, KT, Kinv):
    path = os.path.join(REF_DIR, "libref_flow_%s.so" % which)
    if not os.path.exists(path):
        pytest.skip("oracle/_ref not built (needs /root/reference at build time): %s" % path)
    fn = getattr(ctypes.CDLL(path), SYM)
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4
    B, _, H, W = depth_src.shape
    flow = np.zeros((B, 2, H, W), np.float32)
    valid = np.zeros((B, 1, H, W), np.float32)
    arrs = [np.ascontiguousarray(a, np.float32) for a in (depth_src, depth_tgt, KT, Kinv)]
    fn(flow.ctypes.data, valid.ctypes.data, *[a.ctypes.data for a in arrs], B, H, W, 0)
    return flow, valid


def _cases(small_batch):
    d = small_batch
    KT = oflow.calc_KT(d["src_pose"][0], d["pose_tgt"], d["K"])
    Kinv = np.linalg.inv(d["K"]).astype(np.float32)
    yield "synthetic 480x640 pair", d["depth_rendered"][0], d["depth_gt_observed"], KT, Kinv
    rng = np.random.default_rng(4)
    B, H, W = 3, 37, 53   # odd width, zeros, depths around the 1e-3 cut, reprojections leaving the frame
    src = rng.uniform(0.5, 1.0, (B, 1, H, W)).astype(np.float32)
    src[:, :, :5] = 0
    src[:, :, 5:7] = rng.uniform(5e-4, 2e-3, (B, 1, 2, W)).astype(np.float32)
    tgt = src + rng.normal(0, 2e-3, src.shape).astype(np.float32)
    K = np.array([[60, 0, 26], [0, 60, 18], [0, 0, 1]], np.float32)
    KT2 = np.tile(np.concatenate([K, np.array([[0.9], [0.4], [0.0]], np.float32)], 1), (B, 1, 1)).astype(np.float32)
    yield "random 37x53", src, tgt, KT2, np.linalg.inv(K).astype(np.float32)


def test_flow_bit_exact_vs_reference_kernel_without_contraction(ctx, small_batch):
    for name, src, tgt, KT, Kinv in _cases(small_batch):
        rf, rv = _ref_flow("nofma", src, tgt, KT, Kinv)
        assert rv.sum() > 20, name
        f, v = gpu_flow(src, tgt, KT, Kinv)                       # the `_flow` drop-in (host pointers)
        np.testing.assert_array_equal(v, rv, err_msg=name)
        np.testing.assert_array_equal(f.view(np.uint32), rf.view(np.uint32), err_msg=name)
        B, _, H, W = src.shape                                    # the device-pointer entry
        flow, valid = ctx.empty((B, 2, H, W)), ctx.empty((B, 1, H, W))
        lib.deepim_flow_forward(ctx.handle, flow, valid, ctx.array(src), ctx.array(tgt), ctx.array(KT),
                                np.ascontiguousarray(Kinv), B, H, W)
        np.testing.assert_array_equal(valid.asnumpy(), rv, err_msg=name)
        np.testing.assert_array_equal(flow.asnumpy().view(np.uint32), rf.view(np.uint32), err_msg=name)
        of, ov = oflow.gpu_flow(src, tgt, KT, Kinv)               # and the CPU oracle is pinned by the same build
        np.testing.assert_array_equal(ov, rv, err_msg=name)
        np.testing.assert_array_equal(of.view(np.uint32), rf.view(np.uint32), err_msg=name)


def test_flow_vs_reference_kernel_with_fma_contraction(small_batch):
    for name, src, tgt, KT, Kinv in _cases(small_batch):
        rf, rv = _ref_flow("fma", src, tgt, KT, Kinv)
        f, v = gpu_flow(src, tgt, KT, Kinv)
        flips = float(np.mean(v != rv))
        same = np.broadcast_to(v == rv, (v.shape[0], 1) + v.shape[2:])
        same2 = np.repeat(same, 2, axis=1)
        # flow = projected coordinate - pixel coordinate: compare relative to the coordinate magnitude (≤ 640)
        scale = np.maximum(1.0, np.abs(rf))
        err = float(np.max(np.abs(f - rf)[same2] / np.maximum(scale[same2], 1.0))) if same2.any() else 0.0
        bitdiff = float(np.mean(f.view(np.uint32)[same2] != rf.view(np.uint32)[same2]))
        print("%s: valid flips %.2e, flow max rel diff %.2e, flow words differing %.3f" % (name, flips, err, bitdiff))
        assert flips < 1e-4, (name, flips)
        assert err < 1e-4, (name, err)


def test_flow_through_the_reference_header_binding(ctx, small_batch):
    """B2 as the reference binds it: oracle/_ref/libflow_hpp_client.so is a C++ TU that includes the reference's
    gpu_flow.hpp UNMODIFIED (gpu_flow.pyx:13-16 does the same) and is linked against libdeepim_hip.so — it resolves
    the C++-mangled `_flow`.  Its result must equal the C-linkage `_flow` and deepim_flow_forward bit for bit."""
    path = os.path.join(REF_DIR, "libflow_hpp_client.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libflow_hpp_client.so not built (needs /root/reference at build time)")
    lib.load()
    call = ctypes.CDLL(path).flow_hpp_client_call
    call.restype = None
    call.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4
    for name, src, tgt, KT, Kinv in _cases(small_batch):
        B, _, H, W = src.shape
        flow = np.zeros((B, 2, H, W), np.float32)
        valid = np.zeros((B, 1, H, W), np.float32)
        arrs = [np.ascontiguousarray(a, np.float32) for a in (src, tgt, KT, Kinv)]
        call(flow.ctypes.data, valid.ctypes.data, *[a.ctypes.data for a in arrs], B, H, W, 0)
        assert lib.load().deepim_flow_status() == 0, lib.last_error()
        f, v = gpu_flow(src, tgt, KT, Kinv)
        assert v.sum() > 20, name
        np.testing.assert_array_equal(valid, v, err_msg=name)
        np.testing.assert_array_equal(flow.view(np.uint32), f.view(np.uint32), err_msg=name)
        dflow, dvalid = ctx.empty((B, 2, H, W)), ctx.empty((B, 1, H, W))
        lib.deepim_flow_forward(ctx.handle, dflow, dvalid, ctx.array(src), ctx.array(tgt), ctx.array(KT),
                                np.ascontiguousarray(Kinv), B, H, W)
        np.testing.assert_array_equal(dvalid.asnumpy(), valid, err_msg=name)
        np.testing.assert_array_equal(dflow.asnumpy().view(np.uint32), flow.view(np.uint32), err_msg=name)
