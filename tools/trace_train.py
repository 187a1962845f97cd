#!/usr/bin/env python
"""Dev: kernels of the LAST training iteration of a rocprofv3 --kernel-trace CSV of tools/bench_train.py, in launch order
(duration, gap), split at the markers: first pack_*/sgd kernel = update phase. usage: trace_train.py <kernel_trace.csv>"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*$", "", n)[:64]
names = [short(r["Kernel_Name"]) for r in rows]
# iteration boundaries: the zoom front end of forward_train (bbox_init_kernel) starts an iteration
starts = [i for i, n in enumerate(names) if n.startswith("bbox_init_kernel") and i > 0 and
          (names[i - 1].startswith(("fc_pack", "pack_", "sgd_mom")))]
lo, hi = starts[-2], starts[-1]      # the last COMPLETE iteration
prev = int(rows[lo - 1]["End_Timestamp"])
agg = {}
for r, n in zip(rows[lo:hi], names[lo:hi]):
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%8.1f us gap %6.1f  %s" % ((en - st) / 1e3, (st - prev) / 1e3, n))
    prev = en
