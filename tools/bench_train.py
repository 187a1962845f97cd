"""Dev probe: time one training-style iteration (forward_train + backward + update) and its pieces.
usage: bench_train.py [B] [heads] [step4] [json]   — `heads` adds the refinement decoder with the flow and mask losses; `step4`
times the reference's whole training step instead (module.py:1131-1137: TRAIN_ITER_SIZE = 4 iterations with the device batch
updater — RT_transform, re-render, calc_RT_delta, K·T, lib/flow_c labels, depth > 0.2 mask — between them: net.train_step);
`json` prints one JSON line instead of the sentence (bench.py's other_configs["training_iteration_*" / "training_step_x4_*"])."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mx_deepim_amd import synthetic
from mx_deepim_amd.config import default_config
from mx_deepim_amd.runtime import Context
from mx_deepim_amd.symbols import deepIM_flownet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = Context.get(0)
d = synthetic.make_batch(B, seed=910, n_frames=1)
HEADS = "heads" in sys.argv[2:]
cfg = default_config(); cfg.network.PRED_FLOW = cfg.network.PRED_MASK = HEADS
net = deepIM_flownet().get_symbol(cfg, is_train=True)
net.bind_train(ctx, B, net.init_weights(cfg, seed=91))
gt = (d["depth_gt_observed"] > 0).astype(np.float32)
pco = np.stack([d["pose_tgt"][b][:, :3] @ d["point_cloud_model"][b] + d["pose_tgt"][b][:, 3:4] for b in range(B)]).astype(np.float32)
data = {k: ctx.array(v) for k, v in {"image_observed": d["image_observed"], "image_rendered": d["image_rendered"][0],
        "mask_observed": d["mask_observed"], "mask_rendered": d["mask_rendered"][0], "src_pose": d["src_pose"][0]}.items()}
label = {k: ctx.array(v) for k, v in {"mask_gt_observed": gt, "point_cloud_model": d["point_cloud_model"],
         "point_cloud_weights": np.ones((B, 3, 3000), np.float32), "point_cloud_observed": pco}.items()}
if HEADS:
    from mx_deepim_amd.lib.pair_matching import data_pair
    label["flow"], label["flow_weights"] = data_pair.get_pair_flow(
        {"depth_rendered": ctx.array(d["depth_rendered"][0]), "depth_gt_observed": ctx.array(d["depth_gt_observed"]),
         "pose_rendered": ctx.array(d["src_pose"][0]), "pose_observed": ctx.array(d["pose_tgt"])}, cfg)
STEP4 = "step4" in sys.argv[2:]
if STEP4:
    from mx_deepim_amd.lib.pair_matching.batch_updater_py_multi import batchUpdaterPyMulti
    from mx_deepim_amd.lib.render_glumpy.render_py_multi import Render_Py
    mesh = dict(synthetic.ellipsoid_mesh([0.05, 0.04, 0.035]), texture=synthetic.procedural_texture())
    mesh.pop("colors")
    rm = Render_Py("synthetic", ["ellipsoid"], d["K"], 640, 480, 0.25, 6.0, meshes={"ellipsoid": mesh}, ctx=ctx,
                   pixel_means=synthetic.PIXEL_MEANS[::-1].copy())
    upd = batchUpdaterPyMulti(cfg, 480, 640, render_machine=rm)
    data.update(tgt_pose=ctx.array(d["pose_tgt"]), depth_gt_observed=ctx.array(d["depth_gt_observed"]))
    NIT = cfg.network.TRAIN_ITER_SIZE
    for _ in range(2):
        net.train_step(data, label, upd, lr=1e-6)
    tt, N = ctx.timer(), 5
    tt.start()
    for _ in range(N):
        net.train_step(data, label, upd, lr=1e-6)
    tt.stop()
    step_ms = tt.elapsed_ms() / N
    # the updater alone (its share of the step): NIT - 1 calls per step
    preds = {"rot_est": net.act["rot_norm"], "trans_est": net.act["trans_est"]}
    batch = dict(label); batch.update(data)
    ws = upd.workspace(ctx, B)
    tu = ctx.timer(); tu.start()
    for _ in range(3 * N):
        upd.forward(batch, preds, cfg, out=ws)
    tu.stop()
    upd_ms = tu.elapsed_ms() / (3 * N)
    import json
    rec = {"value": 1e3 / step_ms, "unit": "training steps/s (x%d iterations: forward + backward + SGD step each, device batch updater between them, batch %d)" % (NIT, B),
           "step_ms": step_ms, "iterations_per_s": NIT * 1e3 / step_ms, "pairs_per_s": B * 1e3 / step_ms,
           "batch_updater_ms_per_call": upd_ms, "batch_updater_share": (NIT - 1) * upd_ms / step_ms, "dtype": "f32",
           "workload": "deepim/core/module.py:1131-1137 with TRAIN_ITER_SIZE = %d, %s, 480x640, synthetic pairs, closed loop on the device "
                       "(RT_transform, HIP re-render, calc_RT_delta, K·T + flow labels, mask)" % (
                           NIT, "full graph: encoder + refinement decoder + flow and mask heads + point-matching loss" if HEADS else
                           "pose branch")}
    print(json.dumps(rec) if "json" in sys.argv[2:] else
          "train step x%d B=%d%s: %.2f ms = %.1f steps/s (%.1f iterations/s); batch updater %.3f ms per call = %.1f %% of the step"
          % (NIT, B, " heads" if HEADS else "", step_ms, 1e3 / step_ms, NIT * 1e3 / step_ms, upd_ms, 100 * rec["batch_updater_share"]))
    sys.exit(0)
for _ in range(2):
    net.forward_train(data, label); net.backward(); net.update(1e-6)
ts = [ctx.timer() for _ in range(3)]
N = 5
acc = [0.0, 0.0, 0.0]
for _ in range(N):
    ts[0].start(); net.forward_train(data, label); ts[0].stop()
    ts[1].start(); net.backward(); ts[1].stop()
    ts[2].start(); net.update(1e-6); ts[2].stop()
    for i in range(3): acc[i] += ts[i].elapsed_ms()
fwd, bwd, upd = (a / N for a in acc)
gf = 38.834e9 * B
if "json" in sys.argv[2:]:
    import json
    print(json.dumps({"value": 1e3 / (fwd + bwd + upd), "unit": "training iterations/s (forward + backward + SGD step, batch %d)" % B,
                      "forward_ms": fwd, "backward_ms": bwd, "update_repack_ms": upd, "pairs_per_s": B * 1e3 / (fwd + bwd + upd),
                      "backward_tflops_on_ideal_flops": 2 * gf / bwd / 1e9, "dtype": "f32",
                      "workload": "SURVEY 8f-4: one training-style iteration, %s, 480x640, synthetic pairs" % (
                          "full graph: encoder + refinement decoder + flow and mask heads + point-matching loss" if HEADS else
                          "pose branch: encoder + fc + point-matching loss")}))
    sys.exit(0)
print(("heads " if HEADS else "pose ") + "B=%d: forward %.2f ms (%.0f TF), backward %.2f ms (%.0f TF on 2x forward FLOPs), update+repack %.2f ms; %.1f training iterations/s (pairs/s %.0f)"
      % (B, fwd, gf / fwd / 1e9, bwd, 2 * gf / bwd / 1e9, upd, 1e3 / (fwd + bwd + upd), B * 1e3 / (fwd + bwd + upd)))
