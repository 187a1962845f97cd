"""Size-independent properties at BASELINE's full size (config 2: 16 pairs, 480x640, 4 iterations) — where
the CPU oracle would take minutes, the domain's own invariants check the HIP path bit-for-bit."""
import ctypes

import numpy as np
import pytest

from mx_deepim_amd.config import default_config
from mx_deepim_amd.runtime import lib
from mx_deepim_amd.symbols import deepIM_flownet
from mx_deepim_amd import synthetic

pytestmark = pytest.mark.gpu
B = 16


@pytest.fixture(scope="module")
def batch16():
    return synthetic.make_batch(B, seed=99, n_frames=1)


@pytest.fixture(scope="module")
def net16(ctx):
    cfg = default_config()
    net = deepIM_flownet().get_symbol(cfg)
    params = net.init_weights(cfg, seed=5)
    for k in list(params):          # zero biases → the conv stack is positively homogeneous
        if k.endswith("_bias"):
            params[k] = np.zeros_like(params[k])
    net.bind(ctx, B, params)
    return net


def _data(ctx, d, perm=None):
    sel = (lambda a: a) if perm is None else (lambda a: np.ascontiguousarray(a[perm]))
    return {"image_observed": ctx.array(sel(d["image_observed"])), "image_rendered": ctx.array(sel(d["image_rendered"][0])),
            "mask_observed": ctx.array(sel(d["mask_observed"])), "mask_rendered": ctx.array(sel(d["mask_rendered"][0])),
            "src_pose": ctx.array(sel(d["src_pose"][0]))}


def test_encoder_is_positively_homogeneous_bit_exact(ctx, net16, batch16):
    """conv + LeakyReLU with zero bias: f(0.5 x) == 0.5 f(x) exactly (power-of-two scaling commutes with every
    fp32 rounding), through all 10 layers incl. the split-K ones — any indexing / padding / reduction-order
    slip in the full-size launch geometry breaks this."""
    net16.zoom(_data(ctx, batch16))
    net16.encoder()
    full = net16.act["conv6_1"].asnumpy()
    mid = net16.act["conv3_1"].asnumpy()
    x = net16.act["net_input"].asnumpy()
    net16.act["net_input"].copyfrom(0.5 * x)
    net16.encoder()
    np.testing.assert_array_equal(net16.act["conv3_1"].asnumpy(), 0.5 * mid)
    np.testing.assert_array_equal(net16.act["conv6_1"].asnumpy(), 0.5 * full)
    assert np.abs(full).max() > 1e-3


@pytest.mark.parametrize("streamk", [0, 1], ids=["whole_tile_blocks_exact", "stream_k_default_to_rounding"])
def test_pairs_are_independent_units(ctx, net16, batch16, streamk):
    """Permuting the pairs of the batch permutes every output (no cross-pair term; pixel tiles that straddle image boundaries
    must not leak). Bit for bit when every Winograd tile block is walked whole (wino_streamk = 0); under the default plan the tile
    blocks of a layer's last, partly filled round are cut along K (stream-K), and WHICH tiles those are depends on a pair's place in
    the batch: there the permuted run agrees to the rounding of one more fp32 add per cut — a leak would be an O(1) difference."""
    lib.deepim_set_option(ctx.handle, b"wino_streamk", streamk)
    try:
        ref = net16.refine_iteration(_data(ctx, batch16)).asnumpy()
        se3 = net16.act["se3"].asnumpy()
        perm = np.random.default_rng(0).permutation(B)
        got = net16.refine_iteration(_data(ctx, batch16, perm)).asnumpy()
        se3p = net16.act["se3"].asnumpy()
    finally:
        lib.deepim_set_option(ctx.handle, b"wino_streamk", 1)
    if streamk == 0:
        np.testing.assert_array_equal(se3p, se3[perm])
        np.testing.assert_array_equal(got, ref[perm])
    else:
        assert np.abs(se3p - se3[perm]).max() <= 1e-5 * max(1.0, float(np.abs(se3).max()))
        assert np.abs(got - ref[perm]).max() <= 1e-5 * max(1.0, float(np.abs(ref).max()))


def test_zoom_front_end_invariants(ctx, net16, batch16):
    """Zoomed masks are binary, the rendered object lands inside the frame and fills a sizeable part of it
    (assets/zoom_in.png), background comes out as exactly -mean/255 (black after un-mean)."""
    net16.zoom(_data(ctx, batch16))
    x = net16.act["net_input"].asnumpy()
    zf = net16.act["zoom_factor"].asnumpy()
    assert np.all((x[:, 6:] == 0) | (x[:, 6:] == 1))
    assert np.all(zf[:, 0] == zf[:, 1]) and np.all((zf[:, 0] > 0.02) & (zf[:, 0] < 1.0))
    frac = x[:, 7].mean(axis=(1, 2))
    assert np.all(frac > 0.05) and np.all(frac < 0.9)
    ys, xs = np.nonzero(x[0, 7])
    assert ys.min() > 0 and ys.max() < 479 and xs.min() > 0 and xs.max() < 639
    black = np.float32(-synthetic.PIXEL_MEANS[::-1][0]) / np.float32(255.0)
    assert np.any(x[:, 3] == black)


def test_four_iteration_loop_full_batch_is_deterministic(ctx, net16, batch16):
    outs = []
    for _ in range(2):
        data = _data(ctx, batch16)
        for it in range(4):
            pose = net16.refine_iteration(data).copy()
            data["src_pose"] = pose
        outs.append(pose.asnumpy())
    np.testing.assert_array_equal(outs[0], outs[1])
    assert np.all(np.isfinite(outs[0]))
