"""Dev probe: does running the batch as G micro-batches on G HIP streams hide the per-layer tails?
Times `reps` encoder passes of B pairs as one stream of B vs G streams of B/G (wall clock, synced at both ends)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd.config import default_config  # noqa: E402
from mx_deepim_amd.runtime import Context  # noqa: E402
from mx_deepim_amd.symbols import deepIM_flownet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
cfg = default_config()
FL = 38.834e9


def build(G):
    nets = []
    for g in range(G):
        ctx = Context(0)
        net = deepIM_flownet().get_symbol(cfg)
        net.bind(ctx, a.batch // G, net.init_weights(cfg, seed=1))
        rng = np.random.default_rng(g)
        net.act["net_input"].copyfrom(rng.standard_normal(net.act["net_input"].shape).astype(np.float32))
        net.encoder()       # primes tables / autotune
        ctx.sync()
        nets.append(net)
    return nets


for G in (1, 2, 4):
    nets = build(G)
    for net in nets:
        net.encoder()
    for net in nets:
        net.ctx.sync()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        for net in nets:
            net.encoder()
    for net in nets:
        net.ctx.sync()
    dt = (time.perf_counter() - t0) / a.reps
    print("G=%d streams x B=%d: %.3f ms per %d-pair encoder pass, %.1f TFLOP/s" % (G, a.batch // G, dt * 1e3, a.batch,
                                                                                 FL * a.batch / dt / 1e12))
    del nets
