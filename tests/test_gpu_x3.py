"""Split-fp16 ("x3") conv path on the GPU: values travel as fp16 pairs (hi, lo), products are hi·hi + hi·lo + lo·hi on the fp16
matrix cores with fp32 accumulation. It must be fp32-GRADE: every layer within 1e-5 of the tensor maximum of the fp32 oracle
(the bar the split-K fp32 kernels are held to; observed ~1e-6), se3 / pose within north_star's 1e-4 (observed ~1e-6) — two
orders of magnitude tighter than the plain fp16 path (tests/test_gpu_fp16.py)."""
import ctypes

import numpy as np
import pytest

from oracle import net as onet
from oracle import pipeline as opipe
from mx_deepim_amd.config import default_config
from mx_deepim_amd.runtime import DeviceArray, lib
from mx_deepim_amd.symbols import deepIM_flownet
from mx_deepim_amd import synthetic

pytestmark = pytest.mark.gpu
cf = ctypes.c_float
MEANS_REV = np.ascontiguousarray(synthetic.PIXEL_MEANS[::-1])


def _wscale(w):
    return 2.0 ** int(np.floor(np.log2(1536.0 / np.abs(w).max())))


def _conv_x3(ctx, x, w, b, s, p, slope, sa=16.0, max_split=0):
    B, cin, H, W = x.shape
    cout, _, k, _ = w.shape
    h = ctx.handle
    xs = ctx.empty((B, H, W, 2 * cin), dtype=np.float16)
    lib.deepim_nchw_f32_to_split16(h, xs, ctx.array(x), B, cin, H, W, cf(sa))
    ws = _wscale(w)
    pk = DeviceArray(ctx, (lib.load().deepim_conv_x3_packed_size(cout, cin, k, k) // 2,), dtype=np.float16)
    lib.deepim_conv_x3_pack_weights(h, pk, ctx.array(w), cout, cin, k, k, cf(ws))
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    os_ = ctx.empty((B, ho, wo, 2 * cout), dtype=np.float16)
    lib.deepim_set_option(h, b"conv_max_split", max_split)
    try:
        lib.deepim_conv2d_x3_forward(h, os_, xs, pk, ctx.array(b), B, cin, H, W, cout, k, k, s, p, cf(slope), cf(1.0 / (sa * ws)),
                                     cf(sa))
    finally:
        lib.deepim_set_option(h, b"conv_max_split", 0)
    out = ctx.empty((B, cout, ho, wo))
    lib.deepim_split16_to_nchw_f32(h, out, os_, B, cout, ho, wo, cf(1.0 / sa))
    return out.asnumpy()


def test_split16_round_trip(ctx):
    """hi + lo carries 22 significand bits of value·scale over fp16's normal range; saturation instead of inf."""
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((2, 32, 5, 7)) * np.exp(rng.uniform(-6, 4, (2, 32, 5, 7)))).astype(np.float32)
    x[0, 0, 0, 0], x[0, 1, 0, 0], x[0, 2, 0, 0] = 0.0, 1e9, -1e9
    xs = ctx.empty((2, 5, 7, 64), dtype=np.float16)
    lib.deepim_nchw_f32_to_split16(ctx.handle, xs, ctx.array(x), 2, 32, 5, 7, cf(16.0))
    rec = xs.asnumpy().reshape(2, 5, 7, 2, 2, 16)                     # [..., group, hi/lo, 16]
    hi = rec[..., 0, :].reshape(2, 5, 7, 32).transpose(0, 3, 1, 2).astype(np.float64)
    lo = rec[..., 1, :].reshape(2, 5, 7, 32).transpose(0, 3, 1, 2).astype(np.float64)
    v = np.clip(x.astype(np.float64) * 16.0, -60000, 60000)
    np.testing.assert_array_equal(hi, v.astype(np.float32).astype(np.float16).astype(np.float64))
    err = np.abs(hi + lo - v)
    assert np.all(err <= np.maximum(np.abs(v) * 2.0 ** -21, 2.0 ** -24))
    back = ctx.empty(x.shape)
    lib.deepim_split16_to_nchw_f32(ctx.handle, back, xs, 2, 32, 5, 7, cf(1.0 / 16.0))
    np.testing.assert_allclose(back.asnumpy(), np.clip(x, -3750, 3750), rtol=2.0 ** -20, atol=2.0 ** -27)
    # the two out-of-range values were clamped, and the status word says so (bit 3); reading clears it
    st = ctypes.c_int(0)
    lib.deepim_zoom_status(ctx.handle, ctypes.byref(st))
    assert st.value & 8
    x[0, 1, 0, 0] = x[0, 2, 0, 0] = 1.0
    lib.deepim_nchw_f32_to_split16(ctx.handle, xs, ctx.array(x), 2, 32, 5, 7, cf(16.0))
    lib.deepim_zoom_status(ctx.handle, ctypes.byref(st))
    assert st.value == 0


def test_conv_x3_flags_saturation_and_is_deterministic(ctx):
    """An output beyond fp16's range after scaling sets status bit 3 (the clamp is not silent); two runs are bit-identical."""
    rng = np.random.default_rng(9)
    B, cin, H, W, cout = 1, 32, 8, 8, 128
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) / 17).astype(np.float32)
    b = np.zeros(cout, np.float32)
    st = ctypes.c_int(0)
    lib.deepim_zoom_status(ctx.handle, ctypes.byref(st))
    a = _conv_x3(ctx, x, w, b, 1, 1, 1.0)
    a2 = _conv_x3(ctx, x, w, b, 1, 1, 1.0)
    np.testing.assert_array_equal(a, a2)
    lib.deepim_zoom_status(ctx.handle, ctypes.byref(st))
    assert st.value == 0
    b[3] = 1e6                                              # 1e6 · 16 > 60000: clamped
    _conv_x3(ctx, x, w, b, 1, 1, 1.0)
    lib.deepim_zoom_status(ctx.handle, ctypes.byref(st))
    assert st.value & 8


X3_CASES = [(1, 64, 60, 80, 128, 5, 2, 2), (2, 128, 30, 40, 256, 5, 2, 2), (2, 256, 15, 20, 256, 3, 1, 1), (1, 256, 30, 40, 512, 3, 2, 1),
            (3, 1024, 8, 10, 1024, 3, 1, 1), (2, 32, 9, 11, 128, 3, 1, 1), (1, 96, 7, 6, 384, 3, 2, 1),
            (4, 32, 120, 160, 256, 3, 1, 1)]   # last: 300 tiles of 256x256 → 256 whole + 44 tail tiles cut into K slices


@pytest.mark.parametrize("case", X3_CASES)
def test_conv_x3_is_fp32_grade(ctx, case):
    B, cin, H, W, cout, k, s, p = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    x = np.where(x > 0, x, 0.1 * x).astype(np.float32)                 # like a LeakyReLU output
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ref = onet.conv2d(x, w, b, s, p, 0.1)
    for max_split in (1, 0):
        got = _conv_x3(ctx, x, w, b, s, p, 0.1, max_split=max_split)
        err = np.abs(got.astype(np.float64) - ref).max() / np.abs(ref).max()
        assert err < 1e-5, (max_split, err)
    # the plain fp16 products alone would be ~1e-3 off: the lo terms are really in the sum
    assert err < 4e-6


@pytest.mark.parametrize("shape", [(2, 64, 128), (1, 50, 68), (3, 16, 64), (1, 480, 640)])
def test_conv1_x3_patch_kernel(ctx, shape):
    """conv1 (8 → 64 channels, 7x7 s2 p3) on the persistent split-fp16 patch kernel vs the fp32 oracle; also the fp32 conv1
    with the split16 epilogue (the fallback for other channel counts)."""
    B, H, W = shape
    rng = np.random.default_rng(H + W)
    x = rng.uniform(-1, 1, (B, 8, H, W)).astype(np.float32)
    x[:, 6:] = (x[:, 6:] > 0.3)                                           # mask channels: exact 0 / 1
    w = (rng.standard_normal((64, 8, 7, 7)) / np.sqrt(8 * 49)).astype(np.float32)
    b = rng.standard_normal(64).astype(np.float32)
    ref = onet.conv2d(x, w, b, 2, 3, 0.1)
    ho, wo = ref.shape[2:]
    h, sa, ws = ctx.handle, 16.0, _wscale(w)
    pk = DeviceArray(ctx, (lib.load().deepim_conv1_x3_packed_size() // 2,), dtype=np.float16)
    lib.deepim_conv1_x3_pack_weights(h, pk, ctx.array(w), cf(ws))
    os_ = ctx.empty((B, ho, wo, 128), dtype=np.float16)
    lib.deepim_conv1_x3_forward(h, os_, ctx.array(x), pk, ctx.array(b), B, H, W, cf(0.1), cf(sa), cf(1.0 / (sa * ws)), cf(sa))
    out = ctx.empty((B, 64, ho, wo))
    lib.deepim_split16_to_nchw_f32(h, out, os_, B, 64, ho, wo, cf(1.0 / sa))
    err = np.abs(out.asnumpy().astype(np.float64) - ref).max() / np.abs(ref).max()
    assert err < 2e-6, err
    # fp32 conv1 + split16 epilogue: the fp32 result carried as pairs
    pk32 = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(64, 8, 7, 7) // 4,))
    lib.deepim_conv_pack_weights(h, pk32, ctx.array(w), 64, 8, 7, 7)
    os2 = ctx.empty((B, ho, wo, 128), dtype=np.float16)
    lib.deepim_conv2d_forward_split16(h, os2, ctx.array(x), pk32, ctx.array(b), B, 8, H, W, 64, 7, 7, 2, 3, cf(0.1), cf(sa))
    out2 = ctx.empty((B, 64, ho, wo))
    lib.deepim_split16_to_nchw_f32(h, out2, os2, B, 64, ho, wo, cf(1.0 / sa))
    assert np.abs(out2.asnumpy().astype(np.float64) - ref).max() / np.abs(ref).max() < 2e-6


def test_conv_x3_small_and_large_magnitudes(ctx):
    """Activations 1e-4 … 1e+2 and weights around 1e-3: pairs stay accurate across fp16's normal range (and below it if the
    matrix cores keep fp16 subnormals); the result is held to 1e-5 of the output maximum."""
    rng = np.random.default_rng(5)
    B, cin, H, W, cout, k = 1, 64, 12, 16, 128, 3
    x = (rng.standard_normal((B, cin, H, W)) * np.exp(rng.uniform(-9, 4, (B, cin, H, W)))).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) * 1e-3 * np.exp(rng.uniform(-4, 0, (cout, cin, k, k)))).astype(np.float32)
    b = np.zeros(cout, np.float32)
    ref = onet.conv2d(x, w, b, 1, 1, 1.0)
    got = _conv_x3(ctx, x, w, b, 1, 1, 1.0)
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-5


@pytest.mark.parametrize("heads", [False, True])
def test_x3_iteration_meets_the_fp32_bar(ctx, heads):
    """One refinement iteration at B = 2 with X3_CONV: every encoder activation within 1e-5 of its maximum of the fp32 oracle,
    se3 and pose within 1e-4 (north_star); with the decoder + heads the flow within 1e-4 and < 1e-4 of the mask pixels flip."""
    B = 2
    d = synthetic.make_batch(B, seed=321, n_frames=1)
    cfg = default_config()
    cfg.network.X3_CONV = True
    if heads:
        cfg.TEST.FAST_TEST = False
    net = deepIM_flownet().get_symbol(cfg)
    params = net.init_weights(cfg, seed=7)
    net.bind(ctx, B, params)
    data_np = {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][0], "mask_observed": d["mask_observed"],
               "mask_rendered": d["mask_rendered"][0], "src_pose": d["src_pose"][0]}
    data = {k: ctx.array(v) for k, v in data_np.items()}
    pose = net.refine_iteration(data).asnumpy()
    ref = opipe.refine_iteration(params, data_np, d["K"], MEANS_REV, cfg.dataset.trans_means, cfg.dataset.trans_stds,
                                 cfg.network.ROT_COORD, heads=heads, normalize_flow=cfg.dataset.NORMALIZE_FLOW)
    np.testing.assert_array_equal(net.act["net_input"].asnumpy(), ref["net_input"])
    for name in [g[0] for g in net.enc_geom]:
        got = net.act["conv6_1"].asnumpy() if name == "conv6_1" else net.activation_nchw(name).asnumpy()
        err = np.abs(got.astype(np.float64) - ref[name]).max() / np.abs(ref[name]).max()
        assert err < 1e-5, (name, err)
    se3 = net.act["se3"].asnumpy()
    assert np.abs(se3 - ref["se3"]).max() <= 1e-4 * max(1.0, np.abs(ref["se3"]).max())
    assert np.abs(pose - ref["pose_est"]).max() <= 1e-4
    if heads:
        f, fr = net.act["flow_est"].asnumpy(), ref["flow_est"]
        assert np.abs(f - fr).max() <= 1e-4 * max(1.0, np.abs(fr).max())
        assert np.mean(net.act["mask_observed_pred"].asnumpy() != ref["mask_observed_pred"]) < 1e-4


def test_conv_x3_more_than_2gib_of_input_runs_as_sub_batches(ctx):
    """conv2's geometry at B = 112: 2.2 GB of split16 input, more than one raw-buffer descriptor addresses (bit 31 of the offset
    is the padding marker) → consecutive sub-batches inside the entry; every sample equals the 4-sample run."""
    rng = np.random.default_rng(11)
    B0, reps, cin, H, W, cout = 4, 28, 64, 240, 320, 128
    x = rng.standard_normal((B0, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, 5, 5)) / 40).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    h, sa, ws = ctx.handle, 16.0, _wscale(w)
    pk = DeviceArray(ctx, (lib.load().deepim_conv_x3_packed_size(cout, cin, 5, 5) // 2,), dtype=np.float16)
    lib.deepim_conv_x3_pack_weights(h, pk, ctx.array(w), cout, cin, 5, 5, cf(ws))
    xs = ctx.empty((B0, H, W, 2 * cin), dtype=np.float16)
    lib.deepim_nchw_f32_to_split16(h, xs, ctx.array(x), B0, cin, H, W, cf(sa))
    big = ctx.empty((B0 * reps, H, W, 2 * cin), dtype=np.float16)
    for r in range(reps):
        big[r * B0:(r + 1) * B0].copyfrom(xs)
    assert big.nbytes > 2 ** 31
    ho, wo = 120, 160
    args = (pk, ctx.array(b))
    small = ctx.empty((B0, ho, wo, 2 * cout), dtype=np.float16)
    lib.deepim_conv2d_x3_forward(h, small, xs, *args, B0, cin, H, W, cout, 5, 5, 2, 2, cf(0.1), cf(1.0 / (sa * ws)), cf(sa))
    out = ctx.empty((B0 * reps, ho, wo, 2 * cout), dtype=np.float16)
    lib.deepim_conv2d_x3_forward(h, out, big, *args, B0 * reps, cin, H, W, cout, 5, 5, 2, 2, cf(0.1), cf(1.0 / (sa * ws)), cf(sa))
    ref = ctx.empty((B0, cout, ho, wo))
    lib.deepim_split16_to_nchw_f32(h, ref, small, B0, cout, ho, wo, cf(1.0 / sa))
    ref = ref.asnumpy()
    got = ctx.empty((B0, cout, ho, wo))
    for r in (0, 13, reps - 1):
        lib.deepim_split16_to_nchw_f32(h, got, out[r * B0:(r + 1) * B0], B0, cout, ho, wo, cf(1.0 / sa))
        assert np.abs(got.asnumpy() - ref).max() <= 1e-5 * np.abs(ref).max()


def test_x3_closed_loop_tracks_the_fp32_loop(ctx):
    """Four closed-loop iterations (refine → re-render → mask update → …, as bench.py runs them) on the same 4 pairs: the final
    poses of the split-fp16 mode stay within 1e-6 of the canonical-order fp32 run — the distance at which the default fp32
    kernels (other summation order, split-K) land as well; rasterisation and the zoom crop do not amplify the difference."""
    from mx_deepim_amd.lib.pair_matching.batch_updater_py_multi import update_test_batch
    from mx_deepim_amd.lib.render_glumpy.render_py_multi import Render_Py
    B = 4
    batch = synthetic.make_batch(B, seed=2333, n_frames=1, with_depth=False)
    mesh = dict(synthetic.ellipsoid_mesh([0.05, 0.04, 0.035]), texture=synthetic.procedural_texture())
    mesh.pop("colors")

    def run(mode):
        cfg = default_config()
        cfg.network.X3_CONV = mode == "x3"
        net = deepIM_flownet().get_symbol(cfg)
        params = net.init_weights(cfg, seed=7)
        params["trans_weight"] = params["trans_weight"] * np.float32(0.02)      # keep the object in frame (as bench.py does)
        params["trans_bias"] = params["trans_bias"] * np.float32(0.02)
        if mode == "canonical":
            net.nc8 = False
        net.bind(ctx, B, params)
        rm = Render_Py("synthetic", ["ellipsoid"], batch["K"], 640, 480, 0.25, 6.0, meshes={"ellipsoid": mesh}, ctx=ctx,
                       pixel_means=synthetic.PIXEL_MEANS[::-1].copy())
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 1 if mode == "canonical" else 0)
        try:
            pose = ctx.array(batch["src_pose"][0])
            data = {"image_observed": ctx.array(batch["image_observed"]), "image_rendered": ctx.array(batch["image_rendered"][0]),
                    "mask_rendered": ctx.array(batch["mask_rendered"][0]),
                    "mask_observed": ctx.array(batch["mask_observed_frames"][0]), "src_pose": pose}
            for it in range(4):
                net.refine_iteration(data, pose)
                if it < 3:
                    data = update_test_batch(cfg, data, rm, pose)
                    data["src_pose"] = pose
            return pose.asnumpy()
        finally:
            lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)

    ref = run("canonical")
    assert np.abs(ref - batch["src_pose"][0]).max() > 1e-3          # the loop moved the poses
    assert np.abs(run("default") - ref).max() < 1e-6
    assert np.abs(run("x3") - ref).max() < 1e-6


def test_x3_and_fp16_plans_do_not_depend_on_the_process_environment():
    """Round 2 read DEEPIM_F16_TN4 / _W8 / _NO_TAIL / _NO_DMA from the environment on the launch path: tiling, hence summation
    order, could differ between two ranks of one job. They are context options now (`f16_dev_flags`): two processes with
    different environments produce bit-identical x3 and fp16 outputs (tail-split, split-K and whole-tile layers)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    outs = []
    for extra in ({}, {"DEEPIM_F16_TN4": "1", "DEEPIM_F16_W8": "1", "DEEPIM_F16_NO_TAIL": "1", "DEEPIM_F16_NO_DMA": "1"}):
        env = {k: v for k, v in os.environ.items() if not k.startswith("DEEPIM_F16")}
        env.update(extra)
        r = subprocess.run([sys.executable, os.path.join(here, "_x3_plan_worker.py")], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([l for l in r.stdout.splitlines() if l.startswith("sha256")][-1])
    assert outs[0] == outs[1], outs
    src = open(os.path.join(os.path.dirname(here), "mx_deepim_amd", "csrc", "conv_f16.hip")).read()
    assert src.count("getenv(") == 1 and 'getenv("DEEPIM_CONV_VERBOSE")' in src        # only the verbose switch is left
