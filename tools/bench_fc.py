"""Dev probe: time fc6 (81920 -> 256) at the bench batch."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, lib
ctx = Context.get(0)
B, I, O = 16, 81920, 256
rng = np.random.default_rng(0)
x = ctx.array(rng.standard_normal((B, I)).astype(np.float32))
w = ctx.array((rng.standard_normal((O, I)) / 300).astype(np.float32))
b = ctx.array(rng.standard_normal(O).astype(np.float32))
out = ctx.empty((B, O))
for s in [int(v) for v in (sys.argv[1:] or ["0"])]:
    if s:
        lib.deepim_set_option(ctx.handle, b"fc_slices", s)
    for _ in range(3):
        lib.deepim_fc_forward(ctx.handle, out, x, w, b, B, I, O, ctypes.c_float(0.1))
    t = ctx.timer(); t.start()
    for _ in range(50):
        lib.deepim_fc_forward(ctx.handle, out, x, w, b, B, I, O, ctypes.c_float(0.1))
    t.stop()
    us = t.elapsed_ms() / 50 * 1e3
    ref = x.asnumpy().astype(np.float64) @ w.asnumpy().astype(np.float64).T + b.asnumpy()
    ref = np.where(ref > 0, ref, 0.1 * ref)
    err = np.abs(out.asnumpy() - ref).max() / np.abs(ref).max()
    print("fc6 slices=%d: %.1f us  (%.2f TB/s of weights)  rel err %.1e" % (s, us, O * I * 4 / us / 1e6, err))
