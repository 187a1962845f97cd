"""Dev probe: per-layer timing of the weight-gradient kernel (LDS-staged vs the round-2 register-fed one) and of the data
gradient (stride-2: four parity-class convs; stride-1: one flipped conv) at the encoder geometries.
usage: bench_wgrad.py [B]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
from mx_deepim_amd.symbols.deepIM_flownet import ENCODER
ctx = Context.get(0)
h = ctx.handle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(0)
hh, ww, cin = 480, 640, 8
cf = ctypes.c_float


def timeit(fn, reps=5):
    fn(); fn()
    t = ctx.timer(); t.start()
    for _ in range(reps):
        fn()
    t.stop()
    return t.elapsed_ms() / reps


tot = {"wg1": 0.0, "wg0": 0.0, "dg": 0.0}
for li, (name, cout, k, s, p) in enumerate(ENCODER):
    ho, wo = (hh + 2 * p - k) // s + 1, (ww + 2 * p - k) // s + 1
    x = ctx.array(rng.standard_normal((B, cin, hh, ww)).astype(np.float32))
    dz = ctx.array(rng.standard_normal((B, cout, ho, wo)).astype(np.float32))
    w = ctx.array((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    dw = ctx.empty((cout, cin, k, k))
    fl = 2.0 * cout * cin * k * k * ho * wo * B
    res = {}
    for mode in (1, 0):
        lib.deepim_set_option(h, b"wgrad_lds", mode)
        res[mode] = timeit(lambda: lib.deepim_conv2d_wgrad(h, dw, x, dz, B, cin, hh, ww, cout, k, k, s, p))
    lib.deepim_set_option(h, b"wgrad_lds", 1)
    res[2] = timeit(lambda: lib.deepim_conv2d_wgrad_tm(h, dw, x, dz, B, cin, hh, ww, cout, k, k, s, p))   # tap-major rows (the graph's)
    line = "%-8s Cin %4d %3dx%3d Cout %4d k%d s%d | wgrad tap-major %.3f ms %5.1f TF | LDS %.3f ms %5.1f TF | reg-fed %.3f ms %5.1f TF" % (
        name, cin, hh, ww, cout, k, s, res[2], fl / res[2] / 1e9, res[1], fl / res[1] / 1e9, res[0], fl / res[0] / 1e9)
    tot["wg1"] += res[1]; tot["wg0"] += res[0]; tot["wg2"] = tot.get("wg2", 0.0) + res[2]
    if li > 0:
        dx = ctx.empty((B, cin, hh, ww))
        wt = ctx.empty((cin * cout * k * k,))
        pk = DeviceArray(ctx, (lib.load().deepim_conv_packed_size(cin, cout, k, k) // 4,))
        cls = ctx.empty((B * cin * ((hh + 1) // 2 + k) * ((ww + 1) // 2 + k),))

        def dgrad():
            if s == 1:
                lib.deepim_conv_pack_dgrad(h, pk, w, cout, cin, k, k, 0, 0, 1, k, k, lib.load().deepim_conv_weight_order(h, B, cout, ho, wo, cin, k, k, 1, k - 1 - p))
                lib.deepim_conv2d_forward(h, dx, dz, pk, None, B, cout, ho, wo, cin, k, k, 1, k - 1 - p, cf(1.0), 0, 0)
                return
            lib.deepim_conv2d_dgrad_s2(h, dx, dz, w, ws2, B, cin, hh, ww, cout, k, p)
        ws2 = DeviceArray(ctx, (max(4, lib.load().deepim_conv_dgrad_s2_packed_size(cout, cin, k, p) // 4),))
        ms = timeit(dgrad)
        tot["dg"] += ms
        line += " | dgrad %.3f ms %5.1f TF" % (ms, fl / ms / 1e9)
        if s == 2:   # class by class (the launches before the grouped plan)
            lib.deepim_set_option(h, b"dgrad_group", 0)
            line += " (class by class %.3f)" % timeit(dgrad)
            lib.deepim_set_option(h, b"dgrad_group", 1)
    print(line)
    hh, ww, cin = ho, wo, cout
print("totals B=%d: wgrad tap-major %.2f ms, LDS %.2f ms, reg-fed %.2f ms, dgrad %.2f ms" % (B, tot["wg2"], tot["wg1"], tot["wg0"], tot["dg"]))
