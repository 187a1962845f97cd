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
    return n.startswith(("conv_mfma_kernel", "conv_direct_kernel", "conv_nc8_kernel", "conv_wino_kernel", "splitk_reduce", "tail_reduce",
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
    bench = re.search(r'"ms_per_launch_group": ([0-9.]+)', open(a.bench).read()) if a.bench else None
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
                             "source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, B=%d, FETCH doubled per MI355X_MICROARCH.md)" % (a.md, a.batch)}
    json.dump(allj, open(a.json, "w"), indent=1)
    print("\n".join(lines))


ap = argparse.ArgumentParser()
sub = ap.add_subparsers(dest="cmd", required=True)
s = sub.add_parser("stats"); s.add_argument("trace"); s.add_argument("stats"); s.add_argument("bench", nargs="?")
s.add_argument("--iters", type=int, default=20); s.add_argument("--batch", type=int, default=16)
t = sub.add_parser("traffic"); t.add_argument("fetch"); t.add_argument("write"); t.add_argument("--iters", type=int, default=8)
t.add_argument("--batch", type=int, default=16); t.add_argument("--md", required=True); t.add_argument("--json", required=True)
t.add_argument("--note", default=""); t.add_argument("--key", default="", help="JSON key instead of B<batch> (e.g. x3_B32)")
a = ap.parse_args()
{"stats": cmd_stats, "traffic": cmd_traffic}[a.cmd](a)
