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
    line = "%-8s Cin %4d %3dx%3d Cout %4d k%d s%d | wgrad LDS %.3f ms %5.1f TF | reg-fed %.3f ms %5.1f TF" % (
        name, cin, hh, ww, cout, k, s, res[1], fl / res[1] / 1e9, res[0], fl / res[0] / 1e9)
    tot["wg1"] += res[1]; tot["wg0"] += res[0]
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
            for py in range(2):
                for px in range(2):
                    ky0, kx0 = (py + p) % 2, (px + p) % 2
                    nky, nkx = (k - ky0 + 1) // 2, (k - kx0 + 1) // 2
                    cy0, cx0 = (py + p - ky0) // 2, (px + p - kx0) // 2
                    P = max(nky, nkx) - 1
                    lib.deepim_conv_pack_dgrad(h, pk, w, cout, cin, k, k, ky0, kx0, 2, nky, nkx, lib.load().deepim_conv_weight_order(h, B, cout, ho, wo, cin, nky, nkx, 1, P))
                    lib.deepim_conv2d_forward_remap(h, dx, dz, pk, B, cout, ho, wo, cin, nky, nkx, P, cy0 + P - (nky - 1), cx0 + P - (nkx - 1), hh, ww, py, px)
        ms = timeit(dgrad)
        tot["dg"] += ms
        line += " | dgrad %.3f ms %5.1f TF" % (ms, fl / ms / 1e9)
    print(line)
    hh, ww, cin = ho, wo, cout
print("totals B=%d: wgrad LDS %.2f ms, reg-fed %.2f ms, dgrad %.2f ms" % (B, tot["wg1"], tot["wg0"], tot["dg"]))
