"""Dev probe: per-layer timing of the fp16 conv path (NHWC fp16 in/out) at the encoder geometries, on RANDOM operands (zero-filled
operands clock 15-20 % higher: cdna_hip_programming.md §5.4 rule 25), variants interleaved in one process (rule 24).
usage: bench_layers_f16.py [B] [cin0] [flags,flags,...]     — each `flags` = a value of deepim_set_option("f16_dev_flags")
       (0 = default: ping-pong kernel where the grid fills the chip; 16 = round 3's 4-wave kernel everywhere)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.runtime import Context, DeviceArray, lib
from mx_deepim_amd.symbols.deepIM_flownet import ENCODER
ctx = Context.get(0)
for o in os.environ.get("DEEPIM_OPT", "").split(","):      # e.g. DEEPIM_OPT=conv_max_split=4
    if "=" in o:
        lib.deepim_set_option(ctx.handle, o.split("=")[0].encode(), int(o.split("=")[1]))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cin0 = int(sys.argv[2]) if len(sys.argv) > 2 else 8
variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "16"])]
ROUNDS, REPS = 3, 4
rng = np.random.default_rng(0)
h, w, cin = 480, 640, cin0
layers = []
for name, cout, k, s, p in ENCODER:
    cpad = (cin + 7) // 8 * 8
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    n = B * h * w * cpad
    xr = rng.uniform(-1, 1, min(n, 1 << 24)).astype(np.float16)      # random fill, tiled over the tensor
    x = ctx.array(np.resize(xr, n).reshape(B, h, w, cpad), dtype=np.float16)
    wt = ctx.array((rng.standard_normal((cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32))
    pk = DeviceArray(ctx, (lib.load().deepim_conv_f16_packed_size(cout, cpad, k, k) // 2,), dtype=np.float16)
    lib.deepim_conv_f16_pack_weights(ctx.handle, pk, wt, cout, cin, cpad, k, k)
    out = ctx.empty((B, ho, wo, cout), dtype=np.float16)
    bias = ctx.zeros((cout,))
    args = (ctx.handle, out, x, pk, bias, B, cpad, h, w, cout, k, k, s, p, ctypes.c_float(0.1))
    layers.append((name, args, 2.0 * cout * cin * k * k * ho * wo * B, (cin, h, w, cout, k, s)))
    h, w, cin = ho, wo, cout
times = {v: [[] for _ in layers] for v in variants}
for v in variants:                                   # first-call work (tap tables, scratch, attributes)
    lib.deepim_set_option(ctx.handle, b"f16_dev_flags", v)
    for _, args, _, _ in layers:
        lib.deepim_conv2d_f16_forward(*args)
for r in range(ROUNDS):
    for v in variants:
        lib.deepim_set_option(ctx.handle, b"f16_dev_flags", v)
        for li, (_, args, _, _) in enumerate(layers):
            lib.deepim_conv2d_f16_forward(*args)
            t = ctx.timer(); t.start()
            for _ in range(REPS):
                lib.deepim_conv2d_f16_forward(*args)
            t.stop()
            times[v][li].append(t.elapsed_ms() / REPS)
lib.deepim_set_option(ctx.handle, b"f16_dev_flags", 0)
print("B = %d, %d-channel input, random operands; median of %d interleaved rounds x %d reps; columns = f16_dev_flags %s" % (B, cin0, ROUNDS, REPS, variants))
tot = {v: 0.0 for v in variants}
tot_fl = 0.0
for li, (name, _, fl, g) in enumerate(layers):
    row = "%-11s Cin %4d %3dx%3d Cout %4d k%d s%d:" % ((name,) + g)
    for v in variants:
        ms = float(np.median(times[v][li])); tot[v] += ms
        row += "   %.3f ms %5.0f TF" % (ms, fl / ms / 1e9)
    tot_fl += fl
    print(row)
print("encoder:" + "".join("   %.3f ms %5.0f TF" % (tot[v], tot_fl / tot[v] / 1e9) for v in variants))
