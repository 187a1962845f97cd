#!/usr/bin/env python
"""Per-layer timing of the encoder convs (HIP events on the context stream). Dev tool for kernel tuning:
    python tools/bench_layers.py [--batch 16] [--reps 10]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.config import default_config  # noqa: E402
from mx_deepim_amd.runtime import Context  # noqa: E402
from mx_deepim_amd.symbols import deepIM_flownet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--heads", action="store_true")
    ap.add_argument("--swizzle", type=int, default=1)
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--tile256", type=int, default=0)
    ap.add_argument("--direct", type=int, default=0, help="1: LDS-free register-fed conv kernel")
    ap.add_argument("--maxsplit", type=int, default=0)
    ap.add_argument("--nc8", type=int, default=1, help="1: channel-blocked activations between the layers (default)")
    ap.add_argument("--check", action="store_true", help="compare every layer against the LDS kernel (split-K off)")
    a = ap.parse_args()
    ctx = Context.get(0)
    cfg = default_config()
    if a.heads:
        cfg.TEST.FAST_TEST = False
    cfg.network.FP16_CONV = a.fp16
    net = deepIM_flownet().get_symbol(cfg)
    net.bind(ctx, a.batch, net.init_weights(cfg, seed=1))
    net.nc8 = bool(a.nc8)
    from mx_deepim_amd.runtime import lib
    lib.deepim_set_option(ctx.handle, b"conv_xcd_swizzle", a.swizzle)
    lib.deepim_set_option(ctx.handle, b"conv_tile256", a.tile256)
    lib.deepim_set_option(ctx.handle, b"conv_max_split", a.maxsplit)
    if a.check and not a.fp16:
        rng = np.random.default_rng(0)
        net.act["net_input"].copyfrom(rng.standard_normal(net.act["net_input"].shape).astype(np.float32))
        lib.deepim_set_option(ctx.handle, b"conv_max_split", 1)
        lib.deepim_set_option(ctx.handle, b"conv_direct", 0)
        net.encoder()
        ref = {g[0]: net.act[g[0]].asnumpy() for g in net.enc_geom}
        for mode, ms in (("direct, no split", 1), ("default policy", 0)):
            lib.deepim_set_option(ctx.handle, b"conv_direct", 2 if ms else 1)
            lib.deepim_set_option(ctx.handle, b"conv_max_split", ms)
            src = net.act["net_input"]
            for name, cin, h, w, cout, k, s, p in net.enc_geom:
                net._conv(name, src, net.act[name], net.B, cin, h, w, cout, k, s, p, 0.1)
                got = net.act[name].asnumpy()
                err = float(np.abs(got - ref[name]).max() / max(1e-30, np.abs(ref[name]).max()))
                print(json.dumps({"check": name, "mode": mode, "rel_err_vs_lds_kernel": err}))
                net.act[name].copyfrom(ref[name])     # keep the chain on reference inputs
                src = net.act[name]
        lib.deepim_set_option(ctx.handle, b"conv_max_split", a.maxsplit)
    lib.deepim_set_option(ctx.handle, b"conv_direct", a.direct)
    rng = np.random.default_rng(0)
    net.act["net_input"].copyfrom(rng.standard_normal(net.act["net_input"].shape).astype(np.float32))
    net.encoder()
    ctx.sync()
    src = net.act["net_input"]
    tot_ms, tot_fl = 0.0, 0.0
    import ctypes
    if a.fp16:
        src = net.act["net_input_h"]
    for name, cin, h, w, cout, k, s, p in net.enc_geom:
        t = ctx.timer()
        if a.fp16:
            cpad = (cin + 7) // 8 * 8
            run = lambda: lib.deepim_conv2d_f16_forward(ctx.handle, net.act[name + "_h"], src, net.packed_f16[name],
                                                        net.params[name + "_bias"], net.B, cpad, h, w, cout, k, k, s, p,
                                                        ctypes.c_float(0.1))
        else:
            li = [g[0] for g in net.enc_geom].index(name)
            run = lambda: net.encoder_layer(li, src)
        run()
        t.start()
        for _ in range(a.reps):
            run()
        t.stop()
        ms = t.elapsed_ms() / a.reps
        ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        fl = 2.0 * cout * cin * k * k * ho * wo * net.B
        tot_ms += ms
        tot_fl += fl
        print(json.dumps({"layer": name, "ms": round(ms, 4), "tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                          "M": cout, "N": net.B * ho * wo, "K": cin * k * k}))
        src = net.act[name + "_h"] if a.fp16 else net.act[name]
    print(json.dumps({"layer": "ENCODER", "ms": round(tot_ms, 4), "tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2)}))
    if a.heads:
        t = ctx.timer()
        net.decoder(); net.heads()
        t.start()
        for _ in range(a.reps):
            net.decoder(); net.heads()
        t.stop()
        print(json.dumps({"layer": "DECODER+HEADS", "ms": round(t.elapsed_ms() / a.reps, 4)}))


if __name__ == "__main__":
    main()
