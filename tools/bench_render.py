"""Time the HIP rasteriser at the bench geometry (B pairs, 480x640, LINEMOD-sized mesh) with HIP events."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd import synthetic  # noqa: E402
from mx_deepim_amd.runtime import Context  # noqa: E402
from mx_deepim_amd.lib.render_glumpy.render_py_multi import Render_Py  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--reps", type=int, default=50)
ap.add_argument("--lat", type=int, default=48)
args = ap.parse_args()
ctx = Context.get(0)
mesh = dict(synthetic.ellipsoid_mesh([0.05, 0.04, 0.035], args.lat, 2 * args.lat), texture=synthetic.procedural_texture())
mesh.pop("colors")
rm = Render_Py("unused", ["obj"], synthetic.K_LINEMOD, 640, 480, meshes={"obj": mesh}, ctx=ctx,
               pixel_means=synthetic.PIXEL_MEANS[::-1].copy())
rng = np.random.default_rng(0)
poses = ctx.array(np.stack([synthetic.sample_pose_pair(rng)[1] for _ in range(args.batch)]))
img, dep = ctx.empty((args.batch, 3, 480, 640)), ctx.empty((args.batch, 1, 480, 640))
for _ in range(3):
    rm.render_into(img, dep, 0, poses)
t = ctx.timer()
t.start()
for _ in range(args.reps):
    rm.render_into(img, dep, 0, poses)
t.stop()
ms = t.elapsed_ms() / args.reps
out_bytes = args.batch * 480 * 640 * 4 * 4            # image + depth written
zb_bytes = args.batch * 480 * 640 * 8 * 2              # z-buffer cleared + read back
print("render B=%d V=%d F=%d: %.3f ms per batch, %.1f us per pose, %.2f TB/s of (output + z-buffer) traffic"
      % (args.batch, len(mesh["vertices"]), len(mesh["faces"]), ms, 1e3 * ms / args.batch, (out_bytes + zb_bytes) / ms / 1e9))
print("covered pixels per pose:", int((dep.asnumpy() > 0).sum() / args.batch))
