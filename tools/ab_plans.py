"""Dev probe: A/B split plans of single encoder layers in one process (interleaved, many reps)."""
import argparse, ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.config import default_config
from mx_deepim_amd.runtime import Context, lib
from mx_deepim_amd.symbols import deepIM_flownet

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--layers", default="conv2,conv3,conv3_1")
ap.add_argument("--plans", default="1,-2,-3,-4,-5,-6")
ap.add_argument("--slots", type=int, default=1024)
ap.add_argument("--nc8", type=int, default=1)
a = ap.parse_args()
ctx = Context.get(0)
cfg = default_config()
net = deepIM_flownet().get_symbol(cfg)
net.bind(ctx, a.batch, net.init_weights(cfg, seed=1))
net.nc8 = bool(a.nc8)
lib.deepim_set_option(ctx.handle, b"conv_tail_slots", a.slots)
rng = np.random.default_rng(0)
net.act["net_input"].copyfrom(rng.standard_normal(net.act["net_input"].shape).astype(np.float32))
net.encoder(); ctx.sync()
plans = [int(x) for x in a.plans.split(",")]
src = net.act["net_input"]
for li, (name, cin, h, w, cout, k, s, p) in enumerate(net.enc_geom):
    if name in a.layers.split(","):
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        fl = 2.0 * cout * cin * k * k * ho * wo * net.B
        res = {pl: [] for pl in plans}
        for rnd in range(a.rounds):
            for pl in plans:
                lib.deepim_set_option(ctx.handle, b"conv_force_plan", pl)
                run = lambda: net.encoder_layer(li, src)
                run(); t = ctx.timer(); t.start()
                for _ in range(a.reps):
                    run()
                t.stop()
                res[pl].append(fl / (t.elapsed_ms() / a.reps * 1e-3) / 1e12)
        print(name, " ".join("%d:%.1f" % (pl, max(v)) for pl, v in res.items()))
        lib.deepim_set_option(ctx.handle, b"conv_force_plan", 0)
    src = net.act[name]
