"""Z-group oracle pinned against the reference's OWN operator files.

tests/golden/zoom_golden.npz was produced by tests/golden/make_zoom_golden.py, which imports
/root/reference/deepim/operator_py/zoom_*.py unmodified over a numpy-backed fake `mxnet` and runs their
forward/backward methods under NumPy-1.x ("legacy") scalar promotion — the reference's era and THE PARITY
TARGET — and, as a second reading, under this container's NumPy 2.  Everything here is bit-exact.
"""
import hashlib
import os

import numpy as np
import pytest

from oracle import zoom as oz

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "zoom_golden.npz"))
H, W = 480, 640
MEANS = np.array([123.68, 116.779, 103.939], np.float32)
MEANS_REV = np.ascontiguousarray(MEANS[::-1])


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def rect_valid(r, H, W):
    v = np.zeros((H, W), bool)
    if r[0] >= 0:
        v[r[2]:r[3] + 1, r[0]:r[1] + 1] = True
    return v


def test_forward_factor_matches_reference_lines_legacy():
    """zoom_mask.py:47-103 and zoom_image.py:41-98 on 1200 seeded boxes/poses (incl. empty rendered masks)."""
    real, rend, pose, K = G["fac_real"], G["fac_rend"], G["fac_pose"], G["fac_K"]
    got = np.stack([oz.zoom_factor_from_valid(rect_valid(real[i], H, W), rect_valid(rend[i], H, W), pose[i], K, H, W)
                    for i in range(len(real))])
    np.testing.assert_array_equal(bits(got), bits(G["fac_zoom_mask_legacy"]))
    np.testing.assert_array_equal(bits(got[:240]), bits(G["fac_zoom_image_legacy"]))


def test_forward_factor_numpy2_reading_is_the_other_fixture():
    """The same reference lines under NumPy 2 (float32 scalar op Python int stays float32): kept as a second
    fixture; the oracle reproduces it with promotion='numpy2' and it differs from the target in tx/ty only."""
    real, rend, pose, K = G["fac_real"], G["fac_rend"], G["fac_pose"], G["fac_K"]
    got = np.stack([oz.zoom_factor_from_valid(rect_valid(real[i], H, W), rect_valid(rend[i], H, W), pose[i], K, H, W,
                                              promotion="numpy2") for i in range(len(real))])
    np.testing.assert_array_equal(bits(got), bits(G["fac_zoom_mask_np2"]))
    leg, np2 = G["fac_zoom_mask_legacy"], G["fac_zoom_mask_np2"]
    assert np.array_equal(bits(leg[:, :2]), bits(np2[:, :2]))
    assert (bits(leg[:, 2:]) != bits(np2[:, 2:])).any()


def test_inverse_factor_matches_reference_lines_legacy():
    """zoom_flow.py:36-44 and zoom_mask_with_factor.py:43-52 on 1800 factors."""
    zf = G["inv_in"]
    got = np.array([oz.inverse_factor(z, H, W) for z in zf], np.float32)
    np.testing.assert_array_equal(bits(got), bits(G["inv_flow_legacy"]))
    np.testing.assert_array_equal(bits(got), bits(G["inv_mask_legacy"]))
    got2 = np.array([oz.inverse_factor(z, H, W, promotion="numpy2") for z in zf], np.float32)
    np.testing.assert_array_equal(bits(got2), bits(G["inv_flow_np2"]))


def test_source_indices_match_materialised_grid():
    """Crop indices: the oracle's separable taps == floor of the fake's materialised GridGenerator grid."""
    sel = G["idx_sel"]
    for zf, x0, y0 in ((G["fac_zoom_mask_legacy"][sel], G["idx_fwd_x0"], G["idx_fwd_y0"]),
                       (G["inv_flow_legacy"][sel], G["idx_inv_x0"], G["idx_inv_y0"])):
        idx = oz.sample_indices(zf, H, W)   # the oracle (and the kernel) clamp far-outside indices to [-4, n+4]
        np.testing.assert_array_equal(idx[:, 0, 0, :], np.clip(x0, -4, W + 4))
        np.testing.assert_array_equal(idx[:, 1, :, 0], np.clip(y0, -4, H + 4))
        assert ((x0 >= 0) & (x0 < W)).any() and ((y0 >= 0) & (y0 < H)).any()


def _small(k):
    return G["small_%s_legacy" % k]


def test_every_zoom_op_matches_reference_outputs_small_frame():
    """All seven Z ops, full output tensors, 60x80 frame, B = 4 (one empty rendered mask → fallback branch)."""
    K, pose = _small("K"), _small("pose")
    r = oz.zoom_mask(_small("mo"), _small("mgt"), _small("depth_r"), pose, K)
    for got, key in zip(r, ("zm0", "zm1", "zm2", "zf")):
        np.testing.assert_array_equal(bits(got), bits(_small(key)), err_msg=key)
    zf = _small("zf")
    r = oz.zoom_image(_small("io"), _small("ir"), pose, K, MEANS_REV)
    for got, key in zip(r, ("zi0", "zi1", "zi_zf")):
        np.testing.assert_array_equal(bits(got), bits(_small(key)), err_msg=key)
    for hl in (0, 1):
        r = oz.zoom_image_with_factor(zf, _small("io"), _small("ir"), MEANS_REV, bool(hl))
        np.testing.assert_array_equal(bits(r[0]), bits(_small("ziwf0_hl%d" % hl)))
        np.testing.assert_array_equal(bits(r[1]), bits(_small("ziwf1_hl%d" % hl)))
    r = oz.zoom_depth(zf, _small("dobs"), _small("depth_r"))
    np.testing.assert_array_equal(bits(r[0]), bits(_small("zd0")))
    np.testing.assert_array_equal(bits(r[1]), bits(_small("zd1")))
    r = oz.zoom_flow(zf, _small("flow"), _small("wts"), False)
    np.testing.assert_array_equal(bits(r[0]), bits(_small("zflow")))
    np.testing.assert_array_equal(bits(r[1]), bits(_small("zflow_w")))
    (ri,) = oz.zoom_flow(zf, _small("flow"), None, True)
    np.testing.assert_array_equal(bits(ri), bits(_small("zflow_inv")))
    for inv in (0, 1):
        np.testing.assert_array_equal(bits(oz.zoom_mask_with_factor(zf, _small("mask_in"), bool(inv))),
                                      bits(_small("zmwf_inv%d" % inv)))
        np.testing.assert_array_equal(bits(oz.zoom_trans(zf, _small("trans"), bool(inv))), bits(_small("ztrans_inv%d" % inv)))
        for zg in (0, 1):
            np.testing.assert_array_equal(bits(oz.zoom_trans_backward(zf, _small("trans"), bool(inv), bool(zg))),
                                          bits(_small("ztrans_bwd_inv%d_zg%d" % (inv, zg))))


@pytest.mark.parametrize("tag", ["legacy"])
def test_full_size_zoom_mask_and_image_match_reference(tag):
    """480x640, the repo's synthetic pairs (B = 2): ZoomMask outputs (bit-packed) + ZoomImageWithFactor (sha256)."""
    from mx_deepim_amd import synthetic
    d = synthetic.make_batch(2, seed=2333, n_frames=1)
    mo, mr, sp = d["mask_observed"], d["depth_rendered"][0], d["src_pose"][0]
    r0, _, r2, zf = oz.zoom_mask(mo, mo, mr, sp, d["K"])
    np.testing.assert_array_equal(bits(zf), bits(G["full_zf_" + tag]))
    np.testing.assert_array_equal(np.packbits(r0.astype(np.uint8)), G["full_zm0_bits_" + tag])
    np.testing.assert_array_equal(np.packbits(r2.astype(np.uint8)), G["full_zm2_bits_" + tag])
    q0, q1 = oz.zoom_image_with_factor(zf, d["image_observed"], d["image_rendered"][0], np.ascontiguousarray(synthetic.PIXEL_MEANS[::-1]))
    assert hashlib.sha256(q0.tobytes()).digest() == G["full_zi0_sha_" + tag].tobytes()
    assert hashlib.sha256(q1.tobytes()).digest() == G["full_zi1_sha_" + tag].tobytes()
