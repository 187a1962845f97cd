"""Dev probe: the 4-iteration closed loop (re-render + mask update between iterations, as bench.py runs it) on the same B pairs
with the conv stack in fp32 (canonical order, split-K off), fp32 (default kernels), split-fp16 x3 and plain fp16: how far do the
final poses of the modes drift from the canonical fp32 run?"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd import synthetic
from mx_deepim_amd.config import default_config
from mx_deepim_amd.lib.pair_matching.batch_updater_py_multi import update_test_batch
from mx_deepim_amd.lib.render_glumpy.render_py_multi import Render_Py
from mx_deepim_amd.runtime import Context, lib
from mx_deepim_amd.symbols import deepIM_flownet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = Context.get(0)
batch = synthetic.make_batch(B, seed=2333, n_frames=1, with_depth=False)
mesh = dict(synthetic.ellipsoid_mesh([0.05, 0.04, 0.035]), texture=synthetic.procedural_texture())
mesh.pop("colors")


def run(mode):
    cfg = default_config()
    cfg.network.X3_CONV = mode == "x3"
    cfg.network.FP16_CONV = mode == "fp16"
    net = deepIM_flownet().get_symbol(cfg)
    params = net.init_weights(cfg, seed=7)
    params["trans_weight"] = params["trans_weight"] * np.float32(0.02)
    params["trans_bias"] = params["trans_bias"] * np.float32(0.02)
    if mode == "canonical":
        net.nc8 = False
    net.bind(ctx, B, params)
    rm = Render_Py("synthetic", ["ellipsoid"], batch["K"], 640, 480, 0.25, 6.0, meshes={"ellipsoid": mesh}, ctx=ctx,
                   pixel_means=synthetic.PIXEL_MEANS[::-1].copy())
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 1 if mode == "canonical" else 0)
    pose = ctx.array(batch["src_pose"][0])
    data = {"image_observed": ctx.array(batch["image_observed"]), "image_rendered": ctx.array(batch["image_rendered"][0]),
            "mask_rendered": ctx.array(batch["mask_rendered"][0]), "mask_observed": ctx.array(batch["mask_observed_frames"][0]),
            "src_pose": pose}
    poses = []
    for it in range(4):
        net.refine_iteration(data, pose)
        poses.append(pose.asnumpy().copy())
        if it < 3:
            data = update_test_batch(cfg, data, rm, pose)
            data["src_pose"] = pose
    lib.deepim_set_option(ctx.handle, b"conv_max_split", 0)
    return poses


ref = run("canonical")
for mode in ("default", "x3", "fp16"):
    p = run(mode)
    print("%-8s max |pose - canonical fp32| after iteration 1..4: %s" % (mode, "  ".join("%.2e" % np.abs(a - b).max() for a, b in zip(p, ref))))
