"""CPU: the identities behind csrc/wino.hip against the oracle's direct convolution — exact in float64, and inside the layer bar
(1e-5 of the output range) in float32 at the encoder's reduction lengths; plus the layer-selection rule of the C ABI."""
import ctypes

import numpy as np
import pytest

from oracle import net as onet
from oracle import wino


def _direct64(x, w, stride, pad):
    B, Cin, H, W = x.shape
    Cout, _, k, _ = w.shape
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    xp = np.zeros((B, Cin, H + 2 * pad, W + 2 * pad))
    xp[:, :, pad:pad + H, pad:pad + W] = x
    out = np.zeros((B, Cout, Ho, Wo))
    for a in range(k):
        for b in range(k):
            out += np.einsum("oc,bcyx->boyx", w[:, :, a, b].astype(np.float64),
                             xp[:, :, a:a + stride * Ho:stride, b:b + stride * Wo:stride])
    return out


@pytest.mark.parametrize("shape", [(2, 5, 7, 9, 6), (1, 3, 2, 2, 4), (1, 4, 1, 6, 3)])
def test_winograd_identity_is_exact_in_float64(shape):
    B, cin, H, W, cout = shape
    rng = np.random.default_rng(sum(shape))
    x, w = rng.standard_normal((B, cin, H, W)), rng.standard_normal((cout, cin, 3, 3))
    np.testing.assert_allclose(wino.winograd_f2x2_3x3(x, w), _direct64(x, w, 1, 1), rtol=0, atol=1e-12)


@pytest.mark.parametrize("shape", [(2, 3, 8, 12, 5), (1, 2, 2, 2, 3), (1, 8, 6, 4, 2)])
def test_stride2_5x5_equals_3x3_over_the_four_phases(shape):
    B, cin, H, W, cout = shape
    rng = np.random.default_rng(sum(shape) + 1)
    x, w = rng.standard_normal((B, cin, H, W)), rng.standard_normal((cout, cin, 5, 5))
    ref = _direct64(x, w, 2, 2)
    got = _direct64(wino.space_to_depth(x), wino.s2d_weights_5x5(w), 1, 1)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    np.testing.assert_allclose(wino.winograd_f2x2_3x3(wino.space_to_depth(x), wino.s2d_weights_5x5(w)), ref, rtol=0, atol=1e-12)


@pytest.mark.parametrize("cin", [256, 1024])
def test_float32_winograd_stays_inside_the_layer_bar(cin):
    """The kernel's arithmetic (fp32 transforms, fp32 channel sum, U rounded once) on K = 9 * cin products per output: error relative
    to the output range against the oracle's fp32 direct convolution — the 1e-5 bar of tests/test_gpu_wino.py has a decade of room."""
    rng = np.random.default_rng(cin)
    x = rng.standard_normal((1, cin, 4, 6)).astype(np.float32)
    w = (rng.standard_normal((8, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    ref = onet.conv2d(x, w, np.zeros(8, np.float32), 1, 1, 1.0)
    got = wino.winograd_f2x2_3x3(x, w, dtype=np.float32)
    err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    assert err <= 2e-6, err


def test_layer_selection_rule():
    """deepim_conv_wino_preferred[_s2d] is host arithmetic (no device call): which encoder layers take the Winograd kernel per batch."""
    from mx_deepim_amd.runtime import lib
    L = lib.load()
    k3 = {"conv3_1": (256, 60, 80, 256), "conv4_1": (512, 30, 40, 512), "conv5_1": (512, 15, 20, 512), "conv6_1": (1024, 8, 10, 1024)}
    k5 = {"conv2": (64, 240, 320, 128), "conv3": (128, 120, 160, 256)}

    def chosen(B):
        s = {n for n, (ci, h, w, co) in k3.items() if L.deepim_conv_wino_preferred(None, B, ci, h, w, co)}
        return s | {n for n, (ci, h, w, co) in k5.items() if L.deepim_conv_wino_preferred_s2d(None, B, ci, h, w, co)}
    # the shared-transform kernel (Cout % 64 == 0) splits the input channels where the grid is small: it pays from 64 tiles on
    for B in (32, 16, 8, 4):
        assert chosen(B) == set(k3) | set(k5)
    assert chosen(2) == chosen(1) == (set(k3) | set(k5)) - {"conv6_1"}       # 20 tiles per sample
    assert not L.deepim_conv_wino_preferred(None, 2, 256, 60, 80, 96)         # Cout % 64 != 0: the one-wave kernel's rule, 19 x 3 blocks ...
    assert L.deepim_conv_wino_preferred(None, 32, 256, 60, 80, 96)            # ... from 128 blocks of 32 channels x 128 tiles on
    assert not L.deepim_conv_wino_preferred(None, 32, 12, 60, 80, 256) and not L.deepim_conv_wino_preferred(None, 32, 256, 60, 80, 48)
    assert not L.deepim_conv_wino_preferred_s2d(None, 32, 64, 241, 320, 128)          # odd height: no space-to-depth form
    assert L.deepim_conv_wino_packed_size(256, 256) == 256 * 256 * 64
    assert L.deepim_conv_wino_preferred(None, 64, 256, 60, 80, 256) and not L.deepim_conv_wino_preferred(None, 512, 256, 60, 80, 256)   # >= 2 GiB input


def test_f4x4_3x3_in_float32_layer_error():
    """F(4x4, 3x3) is the same convolution in exact arithmetic; carried in float32 its interpolation points +-2 and 1/24 cost an order of
    magnitude over F(2x2, 3x3) on random operands — 1e-5 of a layer's range at 256 input channels, 1.6e-5 at 512. Round 5 killed it on the
    self-imposed every-layer bar of 1e-5; VERDICT r5 item 3 re-judged it on north_star's own bar (the test below)."""
    from oracle import wino
    rng = np.random.default_rng(0)
    errs = {}
    for cin in (256, 512):
        x = rng.standard_normal((1, cin, 12, 16)).astype(np.float32)
        x *= rng.uniform(size=x.shape) > 0.3
        w = (rng.standard_normal((64, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
        ref = wino.winograd_f2x2_3x3(x.astype(np.float64), w.astype(np.float64))
        assert np.abs(wino.winograd_f4x4_3x3(x.astype(np.float64), w.astype(np.float64)) - ref).max() < 1e-12      # the identity
        scale = np.abs(ref).max()
        errs[cin] = (np.abs(wino.winograd_f2x2_3x3(x, w, np.float32) - ref).max() / scale,
                     np.abs(wino.winograd_f4x4_3x3(x, w, np.float32) - ref).max() / scale)
    assert errs[256][0] < 2e-6 and errs[512][0] < 2e-6            # F(2x2): a comfortable factor inside 1e-5
    assert 5e-6 < errs[256][1] < 5e-5 and 1e-5 < errs[512][1] < 5e-5
    assert errs[512][1] > 10 * errs[512][0]


def test_f4x4_3x3_on_conv3_1_and_conv4_1_meets_north_stars_pose_bar():
    """VERDICT r5 item 3, the accuracy half of the keep criterion: one full refinement iteration of two 480x640 pairs with conv3_1 and
    conv4_1 computed through float32 F(4x4, 3x3) (the two layers a kernel would take first), everything else the oracle — against the
    oracle's own iteration. Layers 5.6e-6 / 6.4e-6 of range on the real activations (sparser than random operands), se3 3.5e-7, pose 7e-9:
    two orders inside the 1e-5 bar (ten times north_star's 1e-4). The arithmetic is NOT what stopped F(4x4, 3x3); the register budget of a
    two-waves-per-SIMD kernel is (profiles/r06_wino44.md: 36 positions = 144 accumulator registers per wave)."""
    from oracle import net as onet, pipeline as opipe, se3 as ose3, wino
    from mx_deepim_amd import synthetic
    from mx_deepim_amd.config import default_config
    from mx_deepim_amd.symbols import deepIM_flownet
    onet.build()
    cfg = default_config()
    params = deepIM_flownet().get_symbol(cfg).init_weights(cfg, seed=2333)
    params["trans_weight"] = params["trans_weight"] * np.float32(0.02)      # as bench.py: keeps the object in frame
    params["trans_bias"] = params["trans_bias"] * np.float32(0.02)
    d = synthetic.make_batch(2, seed=2333, n_frames=1)
    means_rev = np.ascontiguousarray(synthetic.PIXEL_MEANS[::-1])
    data = {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][0], "mask_observed": d["mask_observed"],
            "mask_rendered": d["mask_rendered"][0], "src_pose": d["src_pose"][0]}
    ref = opipe.refine_iteration(params, data, d["K"], means_rev, cfg.dataset.trans_means, cfg.dataset.trans_stds, cfg.network.ROT_COORD)
    x, layer_err = ref["net_input"], {}
    for name, s, p in opipe.ENCODER:
        w, b = params[name + "_weight"], params[name + "_bias"]
        if name in ("conv3_1", "conv4_1"):
            y = wino.winograd_f4x4_3x3(x, w, np.float32) + b.reshape(1, -1, 1, 1)
            x = np.where(y > 0, y, y * np.float32(0.1)).astype(np.float32)
            layer_err[name] = float(np.abs(x - ref[name]).max() / np.abs(ref[name]).max())
        else:
            x = onet.conv2d(x, w, b, s, p, 0.1, pair_order=0)
    _, _, se3 = opipe.pose_head(params, x, ref["zoom_factor"])
    pose = np.stack([ose3.RT_transform(np.asarray(data["src_pose"][i], np.float32), se3[i, :4], se3[i, 4:], cfg.dataset.trans_means,
                                       cfg.dataset.trans_stds, cfg.network.ROT_COORD) for i in range(2)])

    def rel(a, b):
        return float(np.abs(np.asarray(a, np.float64) - b).max() / np.abs(b).max())
    print("F(4x4,3x3) on conv3_1 / conv4_1: layers %s, se3 %.2e, pose %.2e" % (layer_err, rel(se3, ref["se3"]), rel(pose, ref["pose_est"])))
    assert max(layer_err.values()) < 2e-5
    assert rel(se3, ref["se3"]) < 1e-5 and rel(pose, ref["pose_est"]) < 1e-5


def _wino_plan(B, cin, H, W, cout, out_nc8=1, s2d=0):
    from mx_deepim_amd.runtime import lib
    plan = (ctypes.c_int * 9)()
    assert lib.load().deepim_conv_wino_plan(None, B, cin, H, W, cout, out_nc8, s2d, plan) == 0      # host arithmetic: no context, no device
    return list(plan)


# the encoder's Winograd layers as the kernel sees them (the 5x5 stride-2 ones over their space-to-depth input): name -> (Cin, H, W, Cout, out_nc8, s2d)
_WINO_LAYERS = {"conv2": (256, 120, 160, 128, 3, 1), "conv3": (512, 60, 80, 256, 1, 1), "conv3_1": (256, 60, 80, 256, 3, 0),
                "conv4_1": (512, 30, 40, 512, 1, 0), "conv5_1": (512, 15, 20, 512, 1, 0), "conv6_1": (1024, 8, 10, 1024, 0, 0)}


def test_launch_plans_of_the_encoder_layers():
    """deepim_conv_wino_plan under the default options: block shape, K split and stream-K per layer and batch size — what
    profiles/r05_winograd.md and DESIGN section 3 describe (wide persistent blocks and stream-K of the last round on the long grids,
    four-wave blocks + K split on the short ones, no stream-K into an NCHW output or with fewer than two whole rounds)."""
    p32 = {n: _wino_plan(32, *g) for n, g in _WINO_LAYERS.items()}
    assert [p32[n][0] for n in ("conv2", "conv3", "conv3_1", "conv4_1", "conv5_1", "conv6_1")] == [1, 1, 1, 1, 2, 2]
    assert all(p32[n][1] == 256 for n in ("conv2", "conv3", "conv3_1", "conv4_1")) and p32["conv5_1"][1] == 512      # persistent grids
    assert p32["conv3"][4:7] == [8, 3, 9] and p32["conv3_1"][4:7] == [16, 6, 9] and p32["conv4_1"][4:7] == [32, 22, 4]   # 9.375 / 4.69 rounds
    assert p32["conv2"][4] == 0                                   # 18.75 rounds: the cut would save less than its pieces cost
    assert p32["conv5_1"][2] == 2 and p32["conv6_1"][2] == 3 and p32["conv6_1"][4] == 0
    for B in (16, 8, 4):
        for n, g in _WINO_LAYERS.items():
            pl = _wino_plan(B, *g)
            assert pl[0] in (1, 2) and pl[1] in (256, 480, 512), (B, n, pl)
            assert pl[4] == 0 or (pl[6] >= 2 and pl[2] == 1), (B, n, pl)          # stream-K: from two whole rounds on, never with a K split
    assert _wino_plan(2, 256, 60, 80, 96) == [-1] * 9             # Cout % 64 != 0: the one-wave kernel, no plan of this kind


def _streamk_pieces(plan, slots):
    """The piece walk of conv_wino8_kernel's persistent blocks (w8_iter_next / w8_run_owner in csrc/wino.hip), restated."""
    _, _, _, ks, G, q, F, grid0, rem = plan
    nlb = slots // 8

    def owner(u):
        big = rem * (q + 1)
        return u // (q + 1) if u < big else rem + (u - big) // q
    out = []
    for b in range(slots):
        lb, xcd = b >> 3, b & 7
        u = lb * q + min(lb, rem)
        uend = u + q + (1 if lb < rem else 0)
        mine = [((vb * nlb + lb) * 8 + xcd, 0, G, -1, 0) for vb in range(F)]
        while u < uend:
            lt = u // G
            g0 = u - lt * G
            g1 = min(G, g0 + uend - u)
            o0, o1 = owner(lt * G), owner(lt * G + G - 1)
            npieces = o1 - o0 + 1
            mine.append(((F * nlb + lt) * 8 + xcd, g0, g1, -1 if npieces == 1 else lb - o0, 0 if npieces == 1 else npieces))
            u += g1 - g0
        out.append(mine)
    return out


@pytest.mark.parametrize("B", [32, 16, 8, 4])
def test_stream_k_piece_walk_covers_every_granule_once(B):
    """Every (tile block, granule) of a stream-K layer is computed by exactly one piece; the pieces of a cut tile block carry the copy
    numbers 0 … n-1 in K order and all know n (the arrival counter's target); no tile block is cut into more copies than the scratch
    holds; no block walks more than its whole rounds + three pieces."""
    seen_any = False
    for name, g in _WINO_LAYERS.items():
        plan = _wino_plan(B, *g)
        G, q, F, grid0 = plan[4], plan[5], plan[6], plan[7]
        if G == 0:
            continue
        seen_any = True
        slots = plan[1]
        assert grid0 % 8 == 0 and slots in (256, 512)
        tiles = {}
        for blk in _streamk_pieces(plan, slots):
            assert len(blk) <= F + 3
            for bid, g0, g1, copy, npieces in blk:
                assert 0 <= bid < grid0 and 0 <= g0 < g1 <= G
                tiles.setdefault(bid, []).append((g0, g1, copy, npieces))
        assert sorted(tiles) == list(range(grid0)), name           # every tile block, padding included
        for bid, ps in tiles.items():
            ps.sort()
            assert ps[0][0] == 0 and ps[-1][1] == G and all(a[1] == b[0] for a, b in zip(ps, ps[1:])), (name, bid, ps)
            if len(ps) == 1:
                assert ps[0][2:] == (-1, 0)
            else:
                assert [c for _, _, c, _ in ps] == list(range(len(ps))) and {n for *_, n in ps} == {len(ps)}, (name, bid, ps)
                assert len(ps) <= -(-G // q) + 1 <= 8
    assert seen_any or B == 4
