"""Dev probe: timeline of tile 0 of the second ping-pong kernel (build: tools/build_variants_f16.sh trace:"-DDI_PP_TRACE=1", run
with DEEPIM_LIB=variants/lib_trace.so): per wave the prologue, K loop and epilogue durations and the phase period (one s_memtime per
phase, written one phase later: no extra waits). usage: pp_trace.py [layer 1..5] [B]"""
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
tr = ctx.zeros((8 * 128,), dtype=np.uint64)
fn = lib.load().deepim_dev_pp_trace
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(tr.ptr)) == 0
args = (ctx.handle, out, x, pk, ctx.zeros((cout,)), B, cin, h, w, cout, k, k, s, p, ctypes.c_float(0.1))
for _ in range(3):
    lib.deepim_conv2d_f16_forward(*args)
ctx.sync()
tm = ctx.timer(); tm.start()
for _ in range(5):
    lib.deepim_conv2d_f16_forward(*args)
tm.stop()
ms = tm.elapsed_ms() / 5
t = tr.asnumpy().reshape(8, 128).astype(np.int64)
nst = min(120, k * k * cin // 32)
print("%s: Cin %d %dx%d Cout %d k%d s%d, B = %d: %.3f ms per call; %d stages per tile; s_memtime ticks" % (name, cin, h, w, cout, k, s, B, ms, k * k * cin // 32))
print("wave  prologue     loop  epilogue    total | phase period: median   min   max  (stages 4..%d)" % (nst - 2))
for wv in range(8):
    a = t[wv, 8:8 + nst]
    per = np.diff(a[4:nst - 1])
    print("%4d  %8d %8d  %8d %8d |                 %6d %5d %5d" % (wv, t[wv, 1] - t[wv, 0], t[wv, 2] - t[wv, 1], t[wv, 3] - t[wv, 2], t[wv, 3] - t[wv, 0],
                                                                     np.median(per), per.min(), per.max()))
print("wave 4 starts its phase %d ticks after wave 0 (median)" % np.median(t[4, 12:8 + nst - 2] - t[0, 12:8 + nst - 2]))
