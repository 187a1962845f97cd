#!/usr/bin/env python
"""Dev: print the kernels of the LAST refinement iteration of a rocprofv3 --kernel-trace CSV in launch order, with each
kernel's duration and the idle gap before it (µs).  usage: tools/trace_iteration.py <kernel_trace.csv> [n_last_iterations]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n)[:70]


marks = [i for i, r in enumerate(rows) if "rt_transform_kernel" in r["Kernel_Name"] or "pose_tail_kernel" in r["Kernel_Name"]]
lo = marks[-nlast - 1] + 1
hi = marks[-1] + 1
tot_k = tot_g = 0.0
prev_end = int(rows[lo - 1]["End_Timestamp"])
for r in rows[lo:hi]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap, dur = (st - prev_end) / 1e3, (en - st) / 1e3
    tot_k += dur
    tot_g += max(gap, 0)
    print("%8.1f us  gap %6.1f  grid %-9s %s" % (dur, gap, r.get("Grid_Size", "?"), short(r["Kernel_Name"])))
    prev_end = en
print("kernels %.1f us, gaps %.1f us, total %.1f us over %d iteration(s)" % (tot_k, tot_g, tot_k + tot_g, nlast))
