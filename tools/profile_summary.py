#!/usr/bin/env python
"""Turn rocprofv3 CSV output of a `bench.py` run into the summaries committed under profiles/.

  kernel stats:  tools/profile_summary.py stats <kernel_trace.csv> <kernel_stats.csv> <bench.log> --iters N > profiles/rNN_bench_kernel_stats_vK.csv
  HBM traffic:   tools/profile_summary.py traffic <fetch counter_collection.csv> <write counter_collection.csv> --iters N \
                     --md profiles/rNN_hbm_traffic.md --json profiles/hbm_traffic.json

`--iters` = pose-refinement iterations (of the whole batch) inside the timed region / the run. The timed region of the
kernel trace is located as the last N occurrences of the pose-update kernel (one per iteration)."""
import argparse
import csv
import json
import re
import sys
from collections import OrderedDict, defaultdict

ENC_GFLOP_PER_PAIR = 38.834012160   # SURVEY §8d, cfg-std (8-channel input)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def is_conv(n):
    return n.startswith(("conv_mfma_kernel", "conv_direct_kernel", "conv_nc8_kernel", "conv_wino_kernel", "conv_wino8_kernel", "conv_wino4_kernel", "conv_wino2_kernel", "wino_reduce_kernel", "splitk_reduce", "tail_reduce",
                         "conv_f16", "conv1_x3", "splitk_x3", "splitk_f16", "tail_f16", "split16_to_nchw", "nchw_to_split16",
                         "_ZN12_GLOBAL__N_123splitk_x3", "_ZN12_GLOBAL__N_122split16_to"))


def cmd_stats(a):
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    marks = [i for i, n in enumerate(names) if n.startswith(("rt_transform_kernel", "pose_tail_kernel"))]
    assert len(marks) >= a.iters, "trace holds fewer iterations than --iters"
    # timed region = from just after the (iters+1)-th last pose update to the last kernel
    first = marks[-a.iters - 1] + 1 if len(marks) > a.iters else 0
    agg = OrderedDict()
    conv_ns = 0
    for r, n in zip(rows[first:], names[first:]):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        c, t = agg.get(n, (0, 0))
        agg[n] = (c + 1, t + d)
        if is_conv(n):
            conv_ns += d
    total = sum(t for _, t in agg.values())
    out = ["# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline --steps %d --warmup 2 --batch %d" % (a.iters // 4, a.batch),
           "# (B=%d/GPU, 4 iters, 480x640, 1x MI355X). Two views:" % a.batch,
           "# (1) timed region only = the last %d pose-refinement iterations of the kernel trace (the full-run --stats table below also" % a.iters,
           "#     contains the priming and warm-up passes and the one-time weight packing);",
           "kernel,calls,total_us,avg_us,pct"]
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.append("%s,%d,%.1f,%.2f,%.2f" % (n.replace(",", ";"), c, t / 1e3, t / 1e3 / c, 100.0 * t / total))
    ms = conv_ns / 1e6 / a.iters
    tf = ENC_GFLOP_PER_PAIR * a.batch / ms
    bench = re.search(r'"ms_per_launch_group":\s*([0-9.]+)', open(a.bench).read()) if a.bench else None
    out.append("# conv launch group (10 conv launches + split-K reduces) per iteration: %.3f ms -> %.1f TFLOP/s%s" % (
        ms, tf, "; bench.py HIP events in the same (profiled) run: %.3f ms -> %.1f TFLOP/s" % (
            float(bench.group(1)), ENC_GFLOP_PER_PAIR * a.batch / float(bench.group(1))) if bench else ""))
    out.append("# (2) full-run --stats table as rocprofv3 wrote it:")
    out += [l.rstrip("\n") for l in open(a.stats)]
    print("\n".join(out))


def cmd_traffic(a):
    def load(path):
        g = defaultdict(lambda: [0, 0.0])
        rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
        marks = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith(("rt_transform_kernel", "pose_tail_kernel"))]
        assert len(marks) > a.iters, "run holds no priming pass before the last --iters iterations"
        for r in rows[marks[-a.iters - 1] + 1:]:       # the last `iters` iterations: priming/autotuning launches excluded
            n = short(r["Kernel_Name"])
            key = "conv kernels + split-K reduces" if is_conv(n) else (
                "zoom front end (bbox, zoom_factor, resample)" if re.match(r"bbox|zoom_factor|resample|zoom_concat", n) else (
                    "re-render + mask update" if re.match(r"project|raster|resolve|depth_to_mask|mask_b", n) else (
                        "fc / pose head / rt_transform" if re.match(r"fc_|pose_head|pose_tail|rt_transform", n) else n)))
            g[key][0] += 1
            g[key][1] += float(r["Counter_Value"])
        return g
    f, w = load(a.fetch), load(a.write)
    lines = ["# HBM traffic of the hot path from PMC counters (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)", "",
             "cmd: `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --no-cpu-baseline --steps 1 --warmup 1 --batch %d` " % a.batch +
             "(and the same with WRITE_SIZE); the last %d pose-refinement iterations of %d pairs each (warm-up + timed step; the "
             "priming pass is cut off at the pose-update kernel). Counter unit = KB (x1024 B)." % (a.iters, a.batch), "",
             "FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-B requests at "
             "64 B for wide coalesced reads; the dword loads of the conv kernels are uncalibrated, so 'corrected' is an upper bound).", "",
             "| kernel group | dispatches | FETCH raw MB/iter | FETCH corrected MB/iter | WRITE MB/iter |", "|---|---|---|---|---|"]
    res = {}
    for k in f:
        fr = f[k][1] * 1024 / a.iters / 1e6
        wr = w.get(k, [0, 0.0])[1] * 1024 / a.iters / 1e6
        lines.append("| %s | %d | %.1f | %.1f | %.1f |" % (k, f[k][0], fr, 2 * fr, wr))
        res[k] = (fr, wr)
    fr, wr = res["conv kernels + split-K reduces"]
    lines += ["", "Conv launch group (one %d-pair iteration): raw %.2f GB, corrected %.2f GB read+write." % (a.batch, (fr + wr) / 1e3, (2 * fr + wr) / 1e3)]
    if a.note:
        lines.append(a.note)
    open(a.md, "w").write("\n".join(lines) + "\n")
    import os
    allj = json.load(open(a.json)) if os.path.exists(a.json) else {}
    allj = {k: v for k, v in allj.items() if k.startswith(("B", "x3_", "wino_"))}          # one entry per mode / per-GPU batch size
    allj[a.key or "B%d" % a.batch] = {"conv_launch_group_bytes_corrected": (2 * fr + wr) * 1e6, "conv_launch_group_bytes_raw": (fr + wr) * 1e6,
                             "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, B=%d, FETCH doubled per MI355X_MICROARCH.md)" % (a.md.replace("gpurun_out/", "profiles/"), a.batch)}   # (the summaries judged are the copies under profiles/)
    json.dump(allj, open(a.json, "w"), indent=1)
    print("\n".join(lines))


# ---- per-kernel roofline table: bench.py reads the JSON this writes, like hbm_traffic.json -------------------
# encoder layers of deepim/symbols/deepIM_flownet.py:63-107 on the 8-channel FAST_TEST input: (name, Cin, H, W, Cout, k, s, p)
ENC_GEOM = [("conv1", 8, 480, 640, 64, 7, 2, 3), ("conv2", 64, 240, 320, 128, 5, 2, 2), ("conv3", 128, 120, 160, 256, 5, 2, 2),
            ("conv3_1", 256, 60, 80, 256, 3, 1, 1), ("conv4", 256, 60, 80, 512, 3, 2, 1), ("conv4_1", 512, 30, 40, 512, 3, 1, 1),
            ("conv5", 512, 30, 40, 512, 3, 2, 1), ("conv5_1", 512, 15, 20, 512, 3, 1, 1), ("conv6", 512, 15, 20, 1024, 3, 2, 1),
            ("conv6_1", 1024, 8, 10, 1024, 3, 1, 1)]
FP32_PEAK_TF, HBM_PEAK_GBS = 157.3, 8000.0


def layer_flops(geom, kernel, batch):
    """(algorithmic, executed) FLOPs of one launch: the Winograd kernels execute 16 multiply-adds per 2x2 output tile, input and output
    channel (49 of the 4 x 16 (position, phase) pairs on the 5x5 stride-2 layers over the space-to-depth input)."""
    name, cin, h, w, cout, k, s, p = geom
    ho, wo = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
    alg = 2.0 * cout * cin * k * k * ho * wo * batch
    if kernel.startswith("conv_wino"):
        pos_ch = (49 if s == 2 else 16) * cin
        return alg, 2.0 * cout * pos_ch * ((ho + 1) // 2) * ((wo + 1) // 2) * batch
    return alg, alg


def iteration_slices(rows, names, iters):
    marks = [i for i, n in enumerate(names) if n.startswith(("rt_transform_kernel", "pose_tail_kernel"))]
    assert len(marks) > iters, "trace holds fewer iterations than --iters (+ a priming pass)"
    return [(marks[-i - 2] + 1, marks[-i - 1] + 1) for i in range(iters)][::-1]


def load_trace(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    return rows, [short(r["Kernel_Name"]) for r in rows]


def hbm_bytes(kernel, occ, batch, H=480, W=640):
    """Algorithmic HBM bytes of one launch of the Z / F / H kernels SURVEY 8(d) lists (what the kernel must read + write once)."""
    px = batch * H * W
    if kernel.startswith("flow_kernel"):          # lib/flow_c/gpu_flow_kernel.cu:32-69: depth_src + depth_tgt in, flow (2) + valid out
        return 20.0 * px, "20 B/px: two depth maps read, flow (2 ch) + valid written"
    if kernel.startswith("resolve_kernel"):       # z-buffer read + re-armed (8 + 8), RGB (12) + depth (4) + mask (4) written
        return 36.0 * px, "36 B/px: 64-bit z-buffer read and re-armed, image (3 ch) + depth + mask_rendered written"
    if kernel.startswith(("upsample16_kernel", "upsample16x4_kernel")):    # Deconvolution k32 s16 + Crop of the 30x40 head output: occurrence 0 = mask (1 ch), 1 = flow (2 ch)
        c = 1 if occ == 0 else 2
        return 4.0 * c * (px + batch * 30 * 40), "%d-channel full-resolution output written, 30x40 input read" % c
    if kernel.startswith("resample4_kernel"):     # ZoomMaskWithFactor (1 ch) / ZoomFlow (2 ch) back to the camera frame: read + write once
        c = 1 if occ == 0 else 2
        return 8.0 * c * px, "%d channel(s) read and written once at 480x640" % c
    if kernel.startswith("conv_fewout"):          # (conv_fewout_kernel / conv_fewout_quad_kernel, one occurrence count) flow6 (1024 ch, 8x10), flow5 (1026, 15x20), mask head (770, 30x40), flow4 head (770, 30x40)
        cin, h, w = [(1024, 8, 10), (1026, 15, 20), (770, 30, 40), (770, 30, 40)][min(occ, 3)]
        return 4.0 * batch * cin * h * w, "input stream: %d channels at %dx%d read once" % (cin, h, w)
    return None, None


def cmd_perkernel(a):
    import os
    out = {"source": "tools/profile_summary.py perkernel: rocprofv3 --kernel-trace of `bench.py --no-cpu-baseline --no-other-configs --verify 0 "
                     "--steps %d --warmup 2 --batch %d` (and --heads / tools/bench_train.py for the H / F kernels), last %d iterations; "
                     "mfma_busy from a separate --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128))" % (a.iters // 4, a.batch, a.iters),
           "batch": a.batch, "peak_tflops": FP32_PEAK_TF, "hbm_peak_gbs": HBM_PEAK_GBS}
    rows, names = load_trace(a.trace)
    acc = [[g[0], None, 0, 0.0, 0.0] for g in ENC_GEOM]       # layer, kernel, launches, conv ns, reduce ns
    for lo, hi in iteration_slices(rows, names, a.iters):
        li = -1
        for r, n in zip(rows[lo:hi], names[lo:hi]):
            if not is_conv(n):
                continue
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            if n.startswith(("splitk_reduce", "tail_reduce", "wino_reduce")):
                acc[li][4] += d
            else:
                li += 1
                acc[li][1], acc[li][2], acc[li][3] = n, acc[li][2] + 1, acc[li][3] + d
        assert li == len(ENC_GEOM) - 1, "an iteration of the trace does not hold the 10 encoder layers"
    busy = {}
    if a.pmc and os.path.exists(a.pmc):
        prow, pnames = load_trace(a.pmc)
        disp = OrderedDict()
        for r in prow:
            d = disp.setdefault(r["Dispatch_Id"], {"name": short(r["Kernel_Name"]), "t": int(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
        seq = sorted(disp.values(), key=lambda d: d["t"])
        marks = [i for i, d in enumerate(seq) if d["name"].startswith(("rt_transform_kernel", "pose_tail_kernel"))]
        li = -1
        for d in seq[marks[-2] + 1:marks[-1] + 1]:            # the last iteration of the PMC run
            if is_conv(d["name"]) and not d["name"].startswith(("splitk_reduce", "tail_reduce", "wino_reduce")):
                li += 1
                busy[ENC_GEOM[li][0]] = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (max(1.0, d.get("GRBM_GUI_ACTIVE", 0.0)) * 128)
    per, tot_ms = [], 0.0
    for geom, (layer, kern, calls, ns, rns) in zip(ENC_GEOM, acc):
        ms, rms = ns / 1e6 / a.iters, rns / 1e6 / a.iters
        alg, ex = layer_flops(geom, kern, a.batch)
        per.append({"layer": layer, "kernel": kern, "calls_per_iter": calls // a.iters, "ms": ms, "split_k_reduce_ms": rms,
                    "algorithmic_tflops": alg / (ms + rms) / 1e9, "executed_tflops": ex / (ms + rms) / 1e9,
                    "frac": ex / (ms + rms) / 1e9 / FP32_PEAK_TF, "mfma_busy": busy.get(layer)})
        tot_ms += ms + rms
    out["per_kernel"] = per
    out["sum_ms"] = tot_ms
    bench = re.search(r'"ms_per_launch_group":\s*([0-9.]+)', open(a.bench).read()) if a.bench else None
    out["bench_ms_per_launch_group_same_run"] = float(bench.group(1)) if bench else None
    hbm = []
    for path, batch, it in ((a.heads_trace, a.batch, a.iters), (a.train_trace, a.train_batch, None)):
        if not path or not os.path.exists(path):
            continue
        hr, hn = load_trace(path)
        if it:
            slices = iteration_slices(hr, hn, min(it, 8))
        else:
            slices = [(0, len(hr))]
        agg = OrderedDict()
        for lo, hi in slices:
            occ = defaultdict(int)
            for r, n in zip(hr[lo:hi], hn[lo:hi]):
                base = re.sub(r"<.*$", "", n)
                if base == "conv_fewout_quad_kernel": base = "conv_fewout_kernel"      # the heads' two kernel forms share the occurrence count
                if base not in (("flow_kernel", "resolve_kernel", "upsample16_kernel", "upsample16x4_kernel", "resample4_kernel", "conv_fewout_kernel") if it
                                else ("flow_kernel",)):     # the training-step trace: lib/flow_c's kernel only (its other launches mix shapes)
                    continue
                o = occ[base] if it else 0
                occ[base] += 1
                c, t = agg.get((n, o), (0, 0))
                agg[(n, o)] = (c + 1, t + int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for (n, o), (c, t) in agg.items():
            us = t / 1e3 / c
            b, what = hbm_bytes(n, o, batch)
            if b is None:
                continue
            hbm.append({"kernel": n, "call": o, "batch": batch, "us": us, "algorithmic_bytes": b, "achieved_gbs": b / us / 1e3,
                        "frac": b / us / 1e3 / HBM_PEAK_GBS, "bytes": what})
    out["roofline_hbm"] = hbm
    json.dump(out, open(a.json, "w"), indent=1)
    print(json.dumps(out, indent=1))


ap = argparse.ArgumentParser()
sub = ap.add_subparsers(dest="cmd", required=True)
s = sub.add_parser("stats"); s.add_argument("trace"); s.add_argument("stats"); s.add_argument("bench", nargs="?")
s.add_argument("--iters", type=int, default=20); s.add_argument("--batch", type=int, default=16)
t = sub.add_parser("traffic"); t.add_argument("fetch"); t.add_argument("write"); t.add_argument("--iters", type=int, default=8)
t.add_argument("--batch", type=int, default=16); t.add_argument("--md", required=True); t.add_argument("--json", required=True)
t.add_argument("--note", default=""); t.add_argument("--key", default="", help="JSON key instead of B<batch> (e.g. x3_B32)")
k = sub.add_parser("perkernel"); k.add_argument("trace"); k.add_argument("bench", nargs="?"); k.add_argument("--pmc", default="")
k.add_argument("--heads-trace", dest="heads_trace", default=""); k.add_argument("--train-trace", dest="train_trace", default="")
k.add_argument("--train-batch", dest="train_batch", type=int, default=4)
k.add_argument("--iters", type=int, default=80); k.add_argument("--batch", type=int, default=32); k.add_argument("--json", required=True)
a = ap.parse_args()
{"stats": cmd_stats, "traffic": cmd_traffic, "perkernel": cmd_perkernel}[a.cmd](a)
