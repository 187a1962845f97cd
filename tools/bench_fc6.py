#!/usr/bin/env python
"""Dev: fc6 (81920 -> 256) at small batch — packed MFMA path (+ its per-step re-pack in training) vs the plain kernel on the raw
weights. usage: bench_fc6.py [B]"""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = Context.get(0); h = ctx.handle; cf = ctypes.c_float
I, O = 81920, 256
rng = np.random.default_rng(0)
x = ctx.array(rng.standard_normal((B, I)).astype(np.float32)); w = ctx.array((rng.standard_normal((O, I)) / 300).astype(np.float32))
b = ctx.array(rng.standard_normal((O,)).astype(np.float32)); y0, y1 = ctx.empty((B, O)), ctx.empty((B, O))
nb = lib.load().deepim_fc_packed_size(O, I)
pk = DeviceArray(ctx, (nb // 4,))
def timeit(fn, n=20):
    for _ in range(3): fn()
    t = ctx.timer(); t.start()
    for _ in range(n): fn()
    t.stop(); return t.elapsed_ms() / n
t_pack = timeit(lambda: lib.deepim_fc_pack_weights(h, pk, w, O, I))
t_packed = timeit(lambda: lib.deepim_fc_forward_packed(h, y0, x, pk, b, B, I, O, cf(0.1)))
t_plain = timeit(lambda: lib.deepim_fc_forward(h, y1, x, w, b, B, I, O, cf(0.1)))
err = np.abs(y0.asnumpy() - y1.asnumpy()).max() / np.abs(y1.asnumpy()).max()
print("fc6 B=%d: pack %.1f us, packed forward %.1f us, plain forward %.1f us, rel diff %.2e" % (B, t_pack * 1e3, t_packed * 1e3, t_plain * 1e3, err))
