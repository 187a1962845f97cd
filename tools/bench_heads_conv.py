"""Dev probe: the few-filter 3x3 heads (flow / mask predictors over 770-1026 channels, csrc/conv.hip conv_fewout_quad_kernel) as a
function of the channel-slice plan: option conv_fewout_blocks (target grid) x conv_fewout_minc (smallest slice).
usage: bench_heads_conv.py [B]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
ctx = Context.get(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cf = ctypes.c_float
rng = np.random.default_rng(0)
LAYERS = [("Convolution3 (flow)", 770, 30, 40, 2), ("mask_conv3", 770, 30, 40, 1), ("Convolution2", 1026, 15, 20, 2), ("Convolution1", 1024, 8, 10, 2)]
PLANS = [(512, 32), (1024, 32), (2048, 32), (4096, 16), (8192, 8)]
for name, cin, H, W, cout in LAYERS:
    x = ctx.array(rng.standard_normal((B, cin, H, W)).astype(np.float32))
    w = ctx.array((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32))
    pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cout, cin, 3, 3) // 4,))
    lib.deepim_conv_pack_weights(ctx.handle, pk, w, cout, cin, 3, 3)
    bias = ctx.array(rng.standard_normal(cout).astype(np.float32))
    out = ctx.empty((B, cout, H, W))
    run = lambda: lib.deepim_conv2d_forward(ctx.handle, out, x, pk, bias, B, cin, H, W, cout, 3, 3, 1, 1, cf(1.0), 0, 0)
    ref = None
    row = []
    for blocks, minc in PLANS:
        try:
            lib.deepim_set_option(ctx.handle, b"conv_fewout_blocks", blocks)
            lib.deepim_set_option(ctx.handle, b"conv_fewout_minc", minc)
        except RuntimeError:      # an older build (A/B through DEEPIM_LIB): its fixed plan, once
            if (blocks, minc) != PLANS[0]:
                continue
        run(); run()
        got = out.asnumpy()
        if ref is None:
            ref = got
        err = float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))
        ts = []
        for _ in range(3):
            t = ctx.timer(); t.start()
            for _ in range(10):
                run()
            t.stop(); ts.append(t.elapsed_ms() / 10)
        ms = float(np.median(ts))
        gb = B * cin * H * W * 4 / 1e9
        row.append("%d/%d: %.1f us %.2f TB/s (d %.0e)" % (blocks, minc, ms * 1e3, gb / ms, err))
    print("%-20s B %2d  %s" % (name, B, " | ".join(row)))
