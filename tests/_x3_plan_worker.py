"""Worker of tests/test_gpu_x3.py::test_x3_and_fp16_plans_do_not_depend_on_the_process_environment: runs the x3 and fp16 conv
kernels on fixed seeded inputs (layers that take the tail split, the split-K and the whole-tile plans) and prints one sha256
over all output bytes. Launched twice with different environments (the tiling switches of round 2 were environment variables)."""
import ctypes
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib  # noqa: E402

cf = ctypes.c_float
ctx = Context.get(0)
h = ctx.handle
sha = hashlib.sha256()
rng = np.random.default_rng(31)
for (B, cin, H, W, cout, k, s, p) in [(4, 32, 120, 160, 256, 3, 1, 1), (2, 256, 15, 20, 512, 3, 2, 1), (3, 1024, 8, 10, 1024, 3, 1, 1),
                                      (2, 64, 60, 80, 128, 5, 2, 2)]:
    x = rng.standard_normal((B, cin, H, W)).astype(np.float32)
    w = (rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    ho, wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    # x3
    xs = ctx.empty((B, H, W, 2 * cin), dtype=np.float16)
    lib.deepim_nchw_f32_to_split16(h, xs, ctx.array(x), B, cin, H, W, cf(16.0))
    pk = DeviceArray(ctx, (lib.load().deepim_conv_x3_packed_size(cout, cin, k, k) // 2,), dtype=np.float16)
    lib.deepim_conv_x3_pack_weights(h, pk, ctx.array(w), cout, cin, k, k, cf(256.0))
    out = ctx.empty((B, ho, wo, 2 * cout), dtype=np.float16)
    lib.deepim_conv2d_x3_forward(h, out, xs, pk, ctx.array(b), B, cin, H, W, cout, k, k, s, p, cf(0.1), cf(1.0 / (16.0 * 256.0)), cf(16.0))
    sha.update(out.asnumpy().tobytes())
    # plain fp16
    xh = ctx.empty((B, H, W, cin), dtype=np.float16)
    lib.deepim_nchw_f32_to_nhwc_f16(h, xh, ctx.array(x), B, cin, H, W, cin)
    pk2 = DeviceArray(ctx, (lib.load().deepim_conv_f16_packed_size(cout, cin, k, k) // 2,), dtype=np.float16)
    lib.deepim_conv_f16_pack_weights(h, pk2, ctx.array(w), cout, cin, cin, k, k)
    oh = ctx.empty((B, ho, wo, cout), dtype=np.float16)
    lib.deepim_conv2d_f16_forward(h, oh, xh, pk2, ctx.array(b), B, cin, H, W, cout, k, k, s, p, cf(0.1))
    sha.update(oh.asnumpy().tobytes())
print("sha256", sha.hexdigest())
