"""Dev probe: per-layer timing of the fp16 conv path (NHWC fp16 in/out) at the encoder geometries."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
from mx_deepim_amd.symbols.deepIM_flownet import ENCODER
ctx = Context.get(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cin0 = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(0)
h, w, cin = 480, 640, cin0
tot_ms, tot_fl = 0.0, 0.0
for name, cout, k, s, p in ENCODER:
    cpad = (cin + 7) // 8 * 8
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    x = ctx.zeros((B, h, w, cpad), dtype=np.float16)
    wt = ctx.array((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    pk = DeviceArray(ctx, (lib.load().deepim_conv_f16_packed_size(cout, cpad, k, k) // 2,), dtype=np.float16)
    lib.deepim_conv_f16_pack_weights(ctx.handle, pk, wt, cout, cin, cpad, k, k)
    out = ctx.empty((B, ho, wo, cout), dtype=np.float16)
    bias = ctx.zeros((cout,))
    args = (ctx.handle, out, x, pk, bias, B, cpad, h, w, cout, k, k, s, p, ctypes.c_float(0.1))
    for _ in range(2):
        lib.deepim_conv2d_f16_forward(*args)
    t = ctx.timer(); t.start()
    for _ in range(5):
        lib.deepim_conv2d_f16_forward(*args)
    t.stop()
    ms = t.elapsed_ms() / 5
    fl = 2.0 * cout * cin * k * k * ho * wo * B
    tot_ms += ms; tot_fl += fl
    print("%-11s Cin %4d %3dx%3d Cout %4d k%d s%d: %.3f ms  %6.0f TFLOP/s" % (name, cin, h, w, cout, k, s, ms, fl / ms / 1e9))
    h, w, cin = ho, wo, cout
print("encoder: %.3f ms  %.0f TFLOP/s" % (tot_ms, tot_fl / tot_ms / 1e9))
