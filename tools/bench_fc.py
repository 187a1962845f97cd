"""Dev probe: time fc6 (81920 -> 256): the split-K GEMV (deepim_fc_forward) and the one-pass MFMA kernel
(deepim_fc_forward_packed) at several batch sizes."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
ctx = Context.get(0)
I, O = 81920, 256
rng = np.random.default_rng(0)
wn = (rng.standard_normal((O, I)) / 300).astype(np.float32)
w = ctx.array(wn)
pk = DeviceArray(ctx, (lib.load().deepim_fc_packed_size(O, I) // 4,))
lib.deepim_fc_pack_weights(ctx.handle, pk, w, O, I)
b = ctx.array(rng.standard_normal(O).astype(np.float32))
for B in [int(v) for v in (sys.argv[1:] or ["32", "16", "4"])]:
    xn = rng.standard_normal((B, I)).astype(np.float32)
    x, out = ctx.array(xn), ctx.empty((B, O))
    ref = xn.astype(np.float64) @ wn.astype(np.float64).T + b.asnumpy()
    ref = np.where(ref > 0, ref, 0.1 * ref)
    for name, fn, wt in (("gemv", lib.deepim_fc_forward, w), ("mfma", lib.deepim_fc_forward_packed, pk)):
        for _ in range(3):
            fn(ctx.handle, out, x, wt, b, B, I, O, ctypes.c_float(0.1))
        t = ctx.timer(); t.start()
        for _ in range(50):
            fn(ctx.handle, out, x, wt, b, B, I, O, ctypes.c_float(0.1))
        t.stop()
        us = t.elapsed_ms() / 50 * 1e3
        err = np.abs(out.asnumpy() - ref).max() / np.abs(ref).max()
        print("fc6 B=%d %s: %.1f us  (%.2f TB/s of weights)  rel err %.1e" % (B, name, us, O * I * 4 / us / 1e6, err))
