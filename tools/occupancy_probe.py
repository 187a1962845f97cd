#!/usr/bin/env python
"""Dev probe: conv time vs grid size (256 / 512 / 768 / 1536 blocks of identical work) to see whether
co-resident blocks overlap on the MFMA pipe."""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib

ctx = Context.get(0)
lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
cin, cout, H, W, k = int(os.environ.get("CIN", 256)), 512, 32, 32, 3
w = ctx.array(np.random.default_rng(0).standard_normal((cout, cin, k, k)).astype(np.float32) * 0.02)
pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cout, cin, k, k) // 4,))
lib.deepim_conv_pack_weights(ctx.handle, pk, w, cout, cin, k, k)
for B in (8, 16, 24, 48, 96):
    x = ctx.array(np.random.default_rng(1).standard_normal((B, cin, H, W)).astype(np.float32))
    out = ctx.empty((B, cout, H, W))
    run = lambda: lib.deepim_conv2d_forward(ctx.handle, out, x, pk, None, B, cin, H, W, cout, k, k, 1, 1, ctypes.c_float(0.1), 0, 0)
    run(); t = ctx.timer(); t.start()
    for _ in range(5): run()
    t.stop(); ms = t.elapsed_ms() / 5
    fl = 2.0 * cout * cin * k * k * H * W * B
    print(json.dumps({"B": B, "blocks": (B * H * W // 128) * (cout // 128), "ms": round(ms, 4), "tflops": round(fl / ms / 1e9, 1)}))
