"""Dev probe: the encoder's 3x3 stride-1 layers on the direct NC8 kernel vs the Winograd kernel, random operands, interleaved.
usage: bench_wino.py [B]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
ctx = Context.get(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cf = ctypes.c_float
rng = np.random.default_rng(0)
LAYERS = [("conv3_1", 256, 60, 80, 256), ("conv4_1", 512, 30, 40, 512), ("conv5_1", 512, 15, 20, 512), ("conv6_1", 1024, 8, 10, 1024)]
if os.environ.get("WINO_CUSTOM"):     # "name,cin,H,W,cout;..." instead of the encoder's 3x3 layers (e.g. the same tiles at two depths: the per-block fixed cost)
    LAYERS = [(f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4])) for f in (e.split(",") for e in os.environ["WINO_CUSTOM"].split(";"))]
ROUNDS, REPS = (1, 1) if os.environ.get('WINO_PROBE') else (3, 5)   # WINO_PROBE: one dispatch run per (layer, kernel) for a counter pass
WIDE0, WIDE1 = (int(v) for v in os.environ.get('WINO_SHAPES', '1,3').split(','))   # block shapes: the timed one, the other one (1 auto, 0 64x64, 3 128x32, 2 64x32 x2)
lib.deepim_set_option(ctx.handle, b'wino_wide', WIDE0)
lib.deepim_set_option(ctx.handle, b'wino_persistent', int(os.environ.get('WINO_PERSIST', '1')))
lib.deepim_set_option(ctx.handle, b'wino_streamk', int(os.environ.get('WINO_STREAMK', '1')))
ONLY = os.environ.get("WINO_LAYERS", "").split(",") if os.environ.get("WINO_LAYERS") else None
for name, cin, H, W, cout in LAYERS:
    if ONLY and name not in ONLY:
        continue
    n = B * cin * H * W
    x = ctx.array(np.resize(rng.standard_normal(min(n, 1 << 24)).astype(np.float32), n).reshape(B, cin // 8, H, W, 8))
    w = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    wd = ctx.array(w)
    pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cout, cin, 3, 3) // 4,))
    lib.deepim_conv_pack_weights(ctx.handle, pk, wd, cout, cin, 3, 3)
    pw = DeviceArray(ctx, (lib.load().deepim_conv_wino_packed_size(cout, cin) // 4,))
    lib.deepim_conv_wino_pack_weights(ctx.handle, pw, wd, cout, cin)
    bias = ctx.array(rng.standard_normal(cout).astype(np.float32))
    o1, o2 = ctx.empty((B, cout, H, W)), ctx.empty((B, cout, H, W))
    out8 = 0 if name == "conv6_1" else 1
    w42 = None
    if hasattr(lib.load(), "deepim_conv_wino42_packed_size") and cout % 64 == 0:      # F(4,3) x F(2,3), csrc/wino42.hip
        p42 = DeviceArray(ctx, (lib.load().deepim_conv_wino42_packed_size(cout, cin) // 4,))
        lib.deepim_conv_wino42_pack_weights(ctx.handle, p42, wd, cout, cin)
        o3 = ctx.empty((B, cout, H, W))
        w42 = lambda: lib.deepim_conv2d_wino42_forward(ctx.handle, o3, x, p42, bias, B, cin, H, W, cout, cf(0.1), out8, 0, 0)
    direct = lambda: lib.deepim_conv2d_forward_ex(ctx.handle, o1, x, pk, bias, B, cin, H, W, cout, 3, 3, 1, 1, cf(0.1), 0, 0, 1, out8)
    wino = lambda: lib.deepim_conv2d_wino_forward(ctx.handle, o2, x, pw, bias, B, cin, H, W, cout, cf(0.1), out8, 0, 0)
    def wino1():   # the other block shape (128 channels x 32 tiles if the default is 64 x 64, and vice versa)
        lib.deepim_set_option(ctx.handle, b"wino_wide", WIDE1); wino(); lib.deepim_set_option(ctx.handle, b"wino_wide", WIDE0)
    direct(); wino()
    a, b = o1.asnumpy(), o2.asnumpy()
    err = float(np.abs(a - b).max() / max(1.0, np.abs(a).max()))
    td, tw, t1 = [], [], []
    for r in range(ROUNDS):
        for fn, acc in ((direct, td), (wino, tw), (wino1, t1)):
            fn()
            t = ctx.timer(); t.start()
            for _ in range(REPS):
                fn()
            t.stop()
            acc.append(t.elapsed_ms() / REPS)
    d, wv, w1 = float(np.median(td)), float(np.median(tw)), float(np.median(t1))
    fl = 2.0 * cout * cin * 9 * H * W * B
    tiles = B * ((H + 1) // 2) * ((W + 1) // 2)
    fle = 2.0 * cout * cin * 16 * tiles
    print("%-8s B %2d: direct %.3f ms %5.1f TF | winograd %.3f ms  %5.1f TF algorithmic, %5.1f TF executed | x%.2f | max diff %.1e of range | other block shape %.3f ms"
          % (name, B, d, fl / d / 1e9, wv, fl / wv / 1e9, fle / wv / 1e9, d / wv, err, w1))
    if w42 is not None:
        w42()
        c = o3.asnumpy()
        e42 = float(np.abs(a - c).max() / max(1.0, np.abs(a).max()))
        t4 = []
        for r in range(ROUNDS):
            w42()
            t = ctx.timer(); t.start()
            for _ in range(REPS):
                w42()
            t.stop(); t4.append(t.elapsed_ms() / REPS)
        m42 = float(np.median(t4))
        f42 = 2.0 * cout * cin * 24 * B * ((H + 3) // 4) * ((W + 1) // 2)
        print("%-8s B %2d: F(4,3)xF(2,3) %.3f ms  %5.1f TF algorithmic, %5.1f TF executed | x%.2f over F(2x2,3x3) | max diff %.1e of range"
              % (name, B, m42, fl / m42 / 1e9, f42 / m42 / 1e9, wv / m42, e42))
# the 5x5 stride-2 layers: direct NC8 kernel vs the Winograd kernel over the space-to-depth input (zero positions skipped / not skipped)
for name, cin, H, W, cout in [("conv2", 64, 240, 320, 128), ("conv3", 128, 120, 160, 256)]:
    if ONLY and name not in ONLY:
        continue
    n = B * cin * H * W
    xn = ctx.array(np.resize(rng.standard_normal(min(n, 1 << 24)).astype(np.float32), n).reshape(B, cin, H, W))
    x8, xs = ctx.empty((B, cin, H, W)), ctx.empty((B, 4 * cin, H // 2, W // 2))
    lib.deepim_relayout_nc8(ctx.handle, x8, xn, B, cin, H * W, 1)
    lib.deepim_relayout_nc8_s2d(ctx.handle, xs, xn, B, cin, H, W, 1)
    del xn
    w = (rng.standard_normal((cout, cin, 5, 5)) / np.sqrt(cin * 25)).astype(np.float32)
    wd = ctx.array(w)
    pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cout, cin, 5, 5) // 4,))
    lib.deepim_conv_pack_weights(ctx.handle, pk, wd, cout, cin, 5, 5)
    pw = DeviceArray(ctx, (lib.load().deepim_conv_wino_packed_size(cout, 4 * cin) // 4,))
    lib.deepim_conv_wino_pack_weights_s2d(ctx.handle, pw, wd, cout, cin)
    bias = ctx.array(rng.standard_normal(cout).astype(np.float32))
    Ho, Wo = H // 2, W // 2
    o1, o2 = ctx.empty((B, cout, Ho, Wo)), ctx.empty((B, cout, Ho, Wo))
    direct = lambda: lib.deepim_conv2d_forward_ex(ctx.handle, o1, x8, pk, bias, B, cin, H, W, cout, 5, 5, 2, 2, cf(0.1), 0, 0, 1, 1)
    wino = lambda: lib.deepim_conv2d_wino_forward_s2d(ctx.handle, o2, xs, pw, bias, B, cin, H, W, cout, cf(0.1), 1, 0, 0)
    def wino_full():   # the other block shape
        lib.deepim_set_option(ctx.handle, b"wino_wide", WIDE1); wino(); lib.deepim_set_option(ctx.handle, b"wino_wide", WIDE0)
    direct(); wino()
    a, b = o1.asnumpy(), o2.asnumpy()
    err = float(np.abs(a - b).max() / max(1.0, np.abs(a).max()))
    ts = {}
    for r in range(ROUNDS):
        for key, fn in (("direct", direct), ("wino", wino), ("full", wino_full)):
            fn()
            t = ctx.timer(); t.start()
            for _ in range(REPS):
                fn()
            t.stop()
            ts.setdefault(key, []).append(t.elapsed_ms() / REPS)
    d, wv, wf = (float(np.median(ts[k])) for k in ("direct", "wino", "full"))
    fl = 2.0 * cout * cin * 25 * Ho * Wo * B
    print("%-8s B %2d: direct %.3f ms %5.1f TF | winograd over space-to-depth %.3f ms  %5.1f TF algorithmic | x%.2f | max diff %.1e of range | other block shape %.3f ms"
          % (name, B, d, fl / d / 1e9, wv, fl / wv / 1e9, d / wv, err, wf))
