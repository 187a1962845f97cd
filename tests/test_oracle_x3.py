"""The split-fp16 ("x3") arithmetic on the CPU (oracle/x3.py, numpy) against float64: the error model behind the GPU mode's
"fp32-grade" claim, independent of any GPU. Dot products of the encoder's lengths with LeakyReLU-like activations."""
import numpy as np
import pytest

from oracle import x3


def _case(K, seed, w_mag=None):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((512, K)).astype(np.float32)
    a = np.where(a > 0, a, 0.1 * a).astype(np.float32)
    w = (rng.standard_normal(K) * (w_mag or 1.0 / np.sqrt(K))).astype(np.float32)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    return a, w, ref, np.sqrt((ref ** 2).mean())


@pytest.mark.parametrize("K", [1600, 2304, 4608, 9216])      # conv2, conv3_1 / conv4, conv4_1 … conv6, conv6_1
def test_x3_dot_products_are_fp32_grade(K):
    a, w, ref, rms = _case(K, K)
    f32_chain = np.zeros(len(ref), np.float32)
    for k in range(K):                                          # a sequential fp32 chain, what an fp32 FMA loop does
        f32_chain = (f32_chain + a[:, k] * w[k]).astype(np.float32)
    e_f32 = np.abs(f32_chain - ref).max() / rms
    e_x3 = np.abs(x3.dot(a, w) - ref).max() / rms
    e_f16 = np.abs(x3.dot(a, w, terms=1) - ref).max() / rms
    assert e_x3 < 5e-6 and e_x3 < 2 * e_f32 + 1e-6, (e_x3, e_f32)      # as good as an fp32 chain
    assert e_f16 > 50 * e_x3                                            # the lo terms carry two more decimal digits


def test_x3_scaling_keeps_small_weights_and_activations_accurate():
    a, w, ref, rms = _case(2304, 3, w_mag=1e-4)                 # weights far below fp16's normal range before scaling
    assert np.abs(x3.dot(a, w) - ref).max() / rms < 5e-6
    a2 = (a * np.float32(1e-3)).astype(np.float32)              # activations 1e-3: pairs still carry > 16 bits
    ref2 = a2.astype(np.float64) @ w.astype(np.float64)
    assert np.abs(x3.dot(a2, w) - ref2).max() / np.sqrt((ref2 ** 2).mean()) < 5e-5
    # even if the matrix cores flushed fp16 subnormals the result would stay inside north_star's 1e-4
    assert np.abs(x3.dot(a, w, flush_subnormals=True) - ref).max() / rms < 1e-4


def test_split_pairs_and_saturation():
    v = np.array([0.0, 1.0, -3.14159274, 1e-3, 123.456, 4000.0, -1e9], np.float32)
    hi, lo = x3.split(v, 16.0)
    x = np.clip(v.astype(np.float64) * 16, -60000, 60000)
    assert np.all(np.abs(hi + lo - x) <= np.maximum(np.abs(x) * 2.0 ** -21, 2.0 ** -24))
    assert hi[5] + lo[5] == 60000 and hi[6] + lo[6] == -60000    # clamped (the kernels flag it in the status word)
    assert x3.weight_scale(np.array([0.03, -0.2], np.float32)) == 4096.0
