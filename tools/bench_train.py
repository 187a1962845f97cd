"""Dev probe: time one training-style iteration (forward_train + backward + update) and its pieces.
usage: bench_train.py [B] [heads] [json]   — `heads` adds the refinement decoder with the flow and mask losses; `json` prints one
JSON line instead of the sentence (bench.py's other_configs["training_iteration_*"])."""
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
