"""F-group parity on the GPU, through the C ABI, against the CPU oracle (bit-exact for F1)."""
import ctypes

import numpy as np
import pytest

from oracle import flow as oflow
from mx_deepim_amd.runtime import lib
from mx_deepim_amd import synthetic

pytestmark = pytest.mark.gpu


def _kt(batch, f=0):
    return oflow.calc_KT(batch["src_pose"][f], batch["pose_tgt"], batch["K"])


def test_flow_forward_bit_exact(ctx, small_batch):
    d = small_batch
    B, _, H, W = d["depth_rendered"][0].shape
    KT = _kt(d)
    Kinv = np.linalg.inv(d["K"]).astype(np.float32)
    ref_flow, ref_valid = oflow.gpu_flow(d["depth_rendered"][0], d["depth_gt_observed"], KT, Kinv)
    flow, valid = ctx.empty((B, 2, H, W)), ctx.empty((B, 1, H, W))
    lib.deepim_flow_forward(ctx.handle, flow, valid, ctx.array(d["depth_rendered"][0]), ctx.array(d["depth_gt_observed"]),
                            ctx.array(KT), np.ascontiguousarray(Kinv), B, H, W)
    assert ref_valid.sum() > 100, "synthetic pair has no visible overlap"
    np.testing.assert_array_equal(valid.asnumpy(), ref_valid)
    np.testing.assert_array_equal(flow.asnumpy(), ref_flow)


def test_flow_host_entry_matches_device_entry(ctx, small_batch):
    """`_flow` (lib/flow_c/gpu_flow.hpp:1-3 drop-in, host pointers) == device-pointer kernel."""
    d = small_batch
    B, _, H, W = d["depth_rendered"][0].shape
    KT = _kt(d)
    Kinv = np.ascontiguousarray(np.linalg.inv(d["K"]).astype(np.float32))
    src = np.ascontiguousarray(d["depth_rendered"][0])
    tgt = np.ascontiguousarray(d["depth_gt_observed"])
    flow = np.zeros((B, 2, H, W), np.float32)
    valid = np.zeros((B, 1, H, W), np.float32)
    dll = lib.load()
    p = lambda a: ctypes.c_void_p(a.ctypes.data)
    dll._flow(p(flow), p(valid), p(src), p(tgt), p(KT), p(Kinv), B, H, W, 0)
    assert dll.deepim_flow_status() == 0
    ref_flow, ref_valid = oflow.gpu_flow(src, tgt, KT, Kinv)
    np.testing.assert_array_equal(flow, ref_flow)
    np.testing.assert_array_equal(valid, ref_valid)


def test_flow_odd_width_and_empty(ctx):
    rng = np.random.default_rng(1)
    B, H, W = 3, 37, 53  # scalar-path (W % 4 != 0)
    src = rng.uniform(0.5, 1.0, (B, 1, H, W)).astype(np.float32)
    src[:, :, :5] = 0
    tgt = src + rng.normal(0, 1e-3, src.shape).astype(np.float32)
    K = np.array([[60, 0, 26], [0, 60, 18], [0, 0, 1]], np.float32)
    KT = np.tile(np.concatenate([K, np.array([[0.4], [0.2], [0.0]], np.float32)], 1), (B, 1, 1)).astype(np.float32)
    Kinv = np.linalg.inv(K).astype(np.float32)
    rf, rv = oflow.gpu_flow(src, tgt, KT, Kinv)
    flow, valid = ctx.empty((B, 2, H, W)), ctx.empty((B, 1, H, W))
    lib.deepim_flow_forward(ctx.handle, flow, valid, ctx.array(src), ctx.array(tgt), ctx.array(KT),
                            np.ascontiguousarray(Kinv), B, H, W)
    np.testing.assert_array_equal(valid.asnumpy(), rv)
    np.testing.assert_array_equal(flow.asnumpy(), rf)
    lib.deepim_flow_forward(ctx.handle, flow, valid, flow, flow, flow, np.ascontiguousarray(Kinv), 0, H, W)  # B=0 no-op


def test_calc_KT_and_flow_updater(ctx, small_batch):
    d = small_batch
    B, _, H, W = d["depth_rendered"][0].shape
    KT_ref = _kt(d)
    KT = ctx.empty((B, 3, 4))
    lib.deepim_calc_KT(ctx.handle, KT, ctx.array(d["src_pose"][0]), ctx.array(d["pose_tgt"]), d["K"], B)
    np.testing.assert_allclose(KT.asnumpy(), KT_ref, rtol=1e-6, atol=1e-6)
    rf, rw = oflow.flow_updater(d["depth_rendered"][0], d["depth_gt_observed"], d["src_pose"][0], d["pose_tgt"], d["K"])
    flow, wts = ctx.empty((B, 2, H, W)), ctx.empty((B, 2, H, W))
    lib.deepim_flow_updater_forward(ctx.handle, flow, wts, ctx.array(d["depth_rendered"][0]),
                                    ctx.array(d["depth_gt_observed"]), ctx.array(d["src_pose"][0]),
                                    ctx.array(d["pose_tgt"]), d["K"], ctypes.c_float(3e-3), 0, B, H, W)
    got_w = wts.asnumpy()
    mism = float(np.mean(got_w != rw))
    assert mism < 1e-4, mism  # threshold ties may flip on the last ulp of K·se3
    same = got_w == rw
    assert np.array_equal(flow.asnumpy()[same], rf[same])


def test_calc_flow_variant(ctx, small_batch):
    d = small_batch
    B, _, H, W = d["depth_rendered"][0].shape
    KT = _kt(d)
    Kinv = np.ascontiguousarray(np.linalg.inv(d["K"]).astype(np.float32))
    flow, vis = ctx.empty((B, H, W, 2)), ctx.empty((B, H, W))
    lib.deepim_calc_flow_forward(ctx.handle, flow, vis, ctx.array(d["depth_rendered"][0]),
                                 ctx.array(d["depth_gt_observed"]), ctx.array(KT), Kinv, ctypes.c_float(3e-3), 0, B, H, W)
    gf, gv = flow.asnumpy(), vis.asnumpy()
    for b in range(B):
        rf, rv = oflow.calc_flow(d["depth_rendered"][0][b, 0], KT[b], Kinv, d["depth_gt_observed"][b, 0])
        assert np.mean(gv[b] != rv) < 1e-4
        same = gv[b] == rv
        np.testing.assert_allclose(gf[b][same], rf[same], rtol=1e-4, atol=1e-4)


def test_depth_to_mask(ctx):
    x = np.random.default_rng(0).uniform(0, 0.4, 10007).astype(np.float32)
    out = ctx.empty(x.shape)
    lib.deepim_depth_to_mask(ctx.handle, out, ctx.array(x), ctypes.c_float(0.2), x.size)
    np.testing.assert_array_equal(out.asnumpy(), (x > np.float32(0.2)).astype(np.float32))
