"""CPU emulation of the split-fp16 ("x3") arithmetic of csrc/conv_f16.hip.  TEST INFRASTRUCTURE ONLY.

Not a restatement of anything in the reference (which is fp32 only): it states, in numpy, what the MI355X kernels compute
in that mode, so that the claim "fp32-grade" can be checked on the CPU against float64:

    hi = f16(clamp(v * s)),  lo = f16(clamp(v * s) - hi)                      (x3_split)
    a . w  ~=  (sum hi_a*hi_w + hi_a*lo_w + lo_a*hi_w) / (s_a * s_w)          (three fp16 MFMAs, fp32 accumulation)

Products of two fp16 numbers are exact in fp32 (11 + 11 significand bits); the accumulation is emulated as fp32 adds of
16-wide partial sums (one 32x32x16 MFMA k-step each) — the hardware's order inside a k-step is not specified, so GPU tests
compare against the float64 oracle with a tolerance, not against this emulation bit for bit.
"""
import numpy as np

f32 = np.float32


def split(v, scale, flush_subnormals=False):
    """-> (hi, lo) as float32 arrays holding fp16 values."""
    x = np.clip((np.asarray(v, f32) * f32(scale)).astype(f32), -60000.0, 60000.0).astype(f32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(f32)).astype(np.float16)
    if flush_subnormals:      # what a matrix core that flushed fp16 denormals would see
        tiny = f32(2.0 ** -14)
        hi = np.where(np.abs(hi.astype(f32)) < tiny, np.float16(0), hi)
        lo = np.where(np.abs(lo.astype(f32)) < tiny, np.float16(0), lo)
    return hi.astype(f32), lo.astype(f32)


def weight_scale(w):
    """The power of two the host picks for a weight tensor (deepIM_flownet.bind): max |w| * s in [768, 1536]."""
    m = float(np.abs(w).max())
    return 2.0 ** int(np.floor(np.log2(1536.0 / m))) if m > 0 else 1.0


def dot(a, w, s_a=16.0, s_w=None, flush_subnormals=False, terms=3):
    """Rows of `a` (N,K) against the vector `w` (K,) in x3 arithmetic -> float32 (N,).  terms=1: plain fp16 operands."""
    a, w = np.asarray(a, f32), np.asarray(w, f32)
    s_w = weight_scale(w) if s_w is None else s_w
    ah, al = split(a, s_a, flush_subnormals)
    wh, wl = split(w, s_w, flush_subnormals)
    acc = np.zeros(a.shape[0], f32)
    for k0 in range(0, a.shape[1], 16):
        sl = slice(k0, k0 + 16)
        part = ah[:, sl].astype(np.float64) @ wh[sl].astype(np.float64)
        if terms == 3:
            part = part + ah[:, sl].astype(np.float64) @ wl[sl].astype(np.float64) + al[:, sl].astype(np.float64) @ wh[sl].astype(np.float64)
        acc = (acc + part.astype(f32)).astype(f32)
    return (acc * f32(1.0 / (s_a * s_w))).astype(f32)
