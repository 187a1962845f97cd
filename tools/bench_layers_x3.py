"""Dev probe: per-layer timing of the split-fp16 (x3) conv path at the encoder geometries (conv2 … conv6_1; conv1 is fp32)."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
from mx_deepim_amd.symbols.deepIM_flownet import ENCODER
ctx = Context.get(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(0)
h, w, cin = 240, 320, 64
tot_ms, tot_fl = 0.0, 0.0
c = ctypes.c_float
for name, cout, k, s, p in ENCODER[1:]:
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    x = ctx.zeros((B, h, w, 2 * cin), dtype=np.float16)
    wt = ctx.array((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    pk = DeviceArray(ctx, (lib.load().deepim_conv_x3_packed_size(cout, cin, k, k) // 2,), dtype=np.float16)
    lib.deepim_conv_x3_pack_weights(ctx.handle, pk, wt, cout, cin, k, k, c(1024.0))
    out = ctx.empty((B, ho, wo, 2 * cout), dtype=np.float16)
    bias = ctx.zeros((cout,))
    args = (ctx.handle, out, x, pk, bias, B, cin, h, w, cout, k, k, s, p, c(0.1), c(1.0 / 16384.0), c(16.0))
    for _ in range(2):
        lib.deepim_conv2d_x3_forward(*args)
    t = ctx.timer(); t.start()
    for _ in range(5):
        lib.deepim_conv2d_x3_forward(*args)
    t.stop()
    ms = t.elapsed_ms() / 5
    fl = 2.0 * cout * cin * k * k * ho * wo * B
    tot_ms += ms; tot_fl += fl
    print("%-11s Cin %4d %3dx%3d Cout %4d k%d s%d: %.3f ms  %6.0f TFLOP/s fp32-equivalent (%.0f executed)"
          % (name, cin, h, w, cout, k, s, ms, fl / ms / 1e9, 3 * fl / ms / 1e9))
    h, w, cin = ho, wo, cout
print("conv2..conv6_1: %.3f ms  %.0f TFLOP/s fp32-equivalent (%.0f executed on the fp16 matrix cores)"
      % (tot_ms, tot_fl / tot_ms / 1e9, 3 * tot_fl / tot_ms / 1e9))
