"""Dev probe: conv1 of the split-fp16 encoder (patch kernel) at 480x640."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
ctx = Context.get(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
c = ctypes.c_float
rng = np.random.default_rng(0)
x = ctx.array(rng.uniform(-1, 1, (B, 8, 480, 640)).astype(np.float32))
w = ctx.array((rng.standard_normal((64, 8, 7, 7)) / 20).astype(np.float32))
pk = DeviceArray(ctx, (lib.load().deepim_conv1_x3_packed_size() // 2,), dtype=np.float16)
lib.deepim_conv1_x3_pack_weights(ctx.handle, pk, w, c(4096.0))
out = ctx.empty((B, 240, 320, 128), dtype=np.float16)
bias = ctx.zeros((64,))
args = (ctx.handle, out, x, pk, bias, B, 480, 640, c(0.1), c(16.0), c(1.0 / 65536.0), c(16.0))
for _ in range(2):
    lib.deepim_conv1_x3_forward(*args)
t = ctx.timer(); t.start()
for _ in range(5):
    lib.deepim_conv1_x3_forward(*args)
t.stop()
ms = t.elapsed_ms() / 5
fl = 2.0 * 64 * 8 * 49 * 240 * 320 * B
print("conv1 x3 patch kernel B=%d: %.3f ms  %.0f TFLOP/s fp32-equivalent (%.0f executed), in+out %.0f GB/s"
      % (B, ms, fl / ms / 1e9, 3 * fl * 50 / 49 / ms / 1e9, (B * 8 * 480 * 640 * 4 + B * 240 * 320 * 256) / ms / 1e6))
# plain fp16 variant (config 5): NHWC fp16 output
lib.deepim_conv1_x3_pack_weights(ctx.handle, pk, w, c(1.0))
outh = ctx.empty((B, 240, 320, 64), dtype=np.float16)
argsh = (ctx.handle, outh, x, pk, bias, B, 480, 640, c(0.1))
for _ in range(2):
    lib.deepim_conv1_f16_forward(*argsh)
t = ctx.timer(); t.start()
for _ in range(5):
    lib.deepim_conv1_f16_forward(*argsh)
t.stop()
ms = t.elapsed_ms() / 5
print("conv1 f16 patch kernel B=%d: %.3f ms  %.0f TFLOP/s, in+out %.0f GB/s"
      % (B, ms, fl / ms / 1e9, (B * 8 * 480 * 640 * 4 + B * 240 * 320 * 128) / ms / 1e6))
