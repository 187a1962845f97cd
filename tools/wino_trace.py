"""Dev probe: s_memtime sums of one mid-grid block of the Winograd kernel (build: SRC=wino tools/build_variants_f16.sh
wtrace:"-DWINO_TRACE=1", run with DEEPIM_LIB=variants/lib_wtrace.so). usage: wino_trace.py [B]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
ctx = Context.get(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
rng = np.random.default_rng(0)
fn = lib.load().deepim_dev_wino_trace
fn.argtypes = [ctypes.c_void_p]
for name, cin, H, W, cout in [("conv3_1", 256, 60, 80, 256), ("conv4_1", 512, 30, 40, 512)]:
    n = B * cin * H * W
    x = ctx.array(np.resize(rng.standard_normal(1 << 22).astype(np.float32), n).reshape(B, cin // 8, H, W, 8))
    wd = ctx.array((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32))
    pw = DeviceArray(ctx, (lib.load().deepim_conv_wino_packed_size(cout, cin) // 4,))
    lib.deepim_conv_wino_pack_weights(ctx.handle, pw, wd, cout, cin)
    out = ctx.empty((B, cout, H, W))
    tr = ctx.zeros((32,), dtype=np.uint64)
    assert fn(ctypes.c_void_p(tr.ptr)) == 0
    for _ in range(3):
        lib.deepim_conv2d_wino_forward(ctx.handle, out, x, pw, None, B, cin, H, W, cout, ctypes.c_float(0.1), 1, 0, 0)
    ctx.sync()
    t = tr.asnumpy().reshape(4, 8).astype(np.int64)
    print("%s B %d: per wave [prologue, body0 sum, body1 sum, epilogue, total] ticks; 32 MFMAs x 64 cycles = 2048 per body" % (name, B))
    for w in range(4):
        nb = int(t[w, 5])
        print("  wave %d: prologue %6d | body0 %.0f / body | body1 %.0f / body | epilogue %6d | total %7d (%d blocks of 8 channels)"
              % (w, t[w, 0], t[w, 1] / max(nb, 1), t[w, 2] / max(nb, 1), t[w, 3], t[w, 4], nb))
