"""Dev probe: per-phase timeline of the ping-pong fp16 kernel (conv_f16_pp_kernel built with -DDI_PP_TRACE=1:
tools/build_variants_f16.sh trace:"-DDI_PP_TRACE=1"; run with DEEPIM_LIB=variants/lib_trace.so). Waves 0 and 4 of tile 0 stamp
s_memtime at the start of a phase's memory segment (a), before its first barrier (b: reads issued, DMA pieces issued, counted wait
done), after the barrier + lgkmcnt(0) (c) and after the last MFMA was issued (d). usage: pp_trace.py [layer index 1..5] [B]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
from mx_deepim_amd.symbols.deepIM_flownet import ENCODER
li = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx = Context.get(0)
h, w, cin = 480, 640, 8
for i, (name, cout, k, s, p) in enumerate(ENCODER):
    if i == li:
        break
    h, w, cin = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1, cout
rng = np.random.default_rng(0)
n = B * h * w * cin
x = ctx.array(np.resize(rng.uniform(-1, 1, 1 << 22).astype(np.float16), n).reshape(B, h, w, cin), dtype=np.float16)
wt = ctx.array((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
pk = DeviceArray(ctx, (lib.load().deepim_conv_f16_packed_size(cout, cin, k, k) // 2,), dtype=np.float16)
lib.deepim_conv_f16_pack_weights(ctx.handle, pk, wt, cout, cin, cin, k, k)
ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
out = ctx.empty((B, ho, wo, cout), dtype=np.float16)
NPH = 160
tr = ctx.zeros((NPH * 8,), dtype=np.uint64)
fn = lib.load().deepim_dev_pp_trace
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(tr.ptr)) == 0
args = (ctx.handle, out, x, pk, ctx.zeros((cout,)), B, cin, h, w, cout, k, k, s, p, ctypes.c_float(0.1))
for _ in range(3):
    lib.deepim_conv2d_f16_forward(*args)
ctx.sync()
t = tr.asnumpy().reshape(NPH, 2, 4).astype(np.int64)
nph = min(NPH, 2 * (k * k * cin // 32))
print("%s: Cin %d %dx%d Cout %d k%d s%d, B = %d; s_memtime ticks (shader clock); wave 0 | wave 4" % (name, cin, h, w, cout, k, s, B))
print("phase    mem  bar+lgkm  mfma  period |   mem  bar+lgkm  mfma  period")
rows = []
for ph in range(4, nph - 4):
    r = []
    for wv in range(2):
        a, b, c, d = t[ph, wv]
        r += [b - a, c - b, d - c, t[ph + 1, wv, 0] - a]
    rows.append(r)
    if ph < 24:
        print("%5d  %5d  %7d  %5d  %6d | %5d  %7d  %5d  %6d" % tuple([ph] + r))
rows = np.array(rows)
print("median %5d  %7d  %5d  %6d | %5d  %7d  %5d  %6d" % tuple(np.median(rows, 0).astype(int)))
print("mean   %5d  %7d  %5d  %6d | %5d  %7d  %5d  %6d" % tuple(rows.mean(0).astype(int)))
print("even-phase median (B pieces issued) %s   odd-phase median (A pieces + counted wait) %s" % (
    np.median(rows[0::2], 0).astype(int).tolist(), np.median(rows[1::2], 0).astype(int).tolist()))
