"""Dev probe: s_memtime stamps of one mid-grid block of the shared-transform Winograd kernel — entry, first step, loop end, exchange done,
exit — per wave (build: tools/build_variants_wino.sh w8trace:"-DW8_TRACE=1", run with DEEPIM_LIB=variants/lib_w8trace.so).
usage: wino8_trace.py [B]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
ctx = Context.get(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(0)
fn = lib.load().deepim_dev_w8_trace
fn.argtypes = [ctypes.c_void_p]
for shape in (3, 2):
    lib.deepim_set_option(ctx.handle, b"wino_wide", shape)
    for name, cin, H, W, cout in [("conv3_1", 256, 60, 80, 256), ("conv4_1", 512, 30, 40, 512)]:
        n = B * cin * H * W
        x = ctx.array(np.resize(rng.standard_normal(1 << 22).astype(np.float32), n).reshape(B, cin // 8, H, W, 8))
        wd = ctx.array((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32))
        pw = DeviceArray(ctx, (lib.load().deepim_conv_wino_packed_size(cout, cin) // 4,))
        lib.deepim_conv_wino_pack_weights(ctx.handle, pw, wd, cout, cin)
        out = ctx.empty((B, cout, H, W))
        tr = ctx.zeros((64,), dtype=np.uint64)
        assert fn(ctypes.c_void_p(tr.ptr)) == 0
        for _ in range(3):
            lib.deepim_conv2d_wino_forward(ctx.handle, out, x, pw, None, B, cin, H, W, cout, ctypes.c_float(0.1), 1, 0, 0)
        ctx.sync()
        t = tr.asnumpy().reshape(8, 8).astype(np.int64)
        print("%s B %d shape %d: per wave ticks [prologue: to stage 0 visible + rest | K loop (%d steps) | output transform, sums sent | + barrier | finish + stores | total]" % (name, B, shape, cin // 8))
        for w in range(8):
            if t[w, 0] == 0:
                continue
            print("  wave %d: %6d + %5d | %7d (%.0f / step) | %6d | %5d | %6d | %7d" % (w, t[w, 5] - t[w, 0], t[w, 1] - t[w, 5], t[w, 2] - t[w, 1], (t[w, 2] - t[w, 1]) / (cin // 8),
                                                                                    t[w, 6] - t[w, 2], t[w, 3] - t[w, 6], t[w, 4] - t[w, 3], t[w, 4] - t[w, 0]))
        t0 = t[t[:, 0] > 0]
        print("  block: first entry -> last exit %d ticks; entry spread %d, exit spread %d" % (t0[:, 4].max() - t0[:, 0].min(), t0[:, 0].max() - t0[:, 0].min(), t0[:, 4].max() - t0[:, 4].min()))
