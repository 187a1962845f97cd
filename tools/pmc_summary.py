#!/usr/bin/env python
"""Dev: per-kernel PMC summary of a rocprofv3 --pmc run (counter_collection.csv). usage: pmc_summary.py <counter_collection.csv>
Prints, per run of consecutive dispatches of one (kernel name, grid) — its last dispatch; grid#run ordinal —: MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)
… (same formula as profiles/r02_pmc_*.md), wait / active fractions of SQ_WAVE_CYCLES, LDS bank conflict share."""
import csv, re, sys
from collections import OrderedDict
rows = list(csv.DictReader(open(sys.argv[1])))
disp = OrderedDict()
for r in rows:
    key = r["Dispatch_Id"]
    d = disp.setdefault(key, {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", "?"), "vgpr": r.get("VGPR_Count", "?"), "agpr": r.get("Accum_VGPR_Count", "?"),
                              "ns": (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) if r.get("End_Timestamp") else 0.0})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
# one row per RUN of consecutive dispatches of the same (name, grid) — the persistent Winograd kernels launch the same grid for every layer, so
# the name + grid alone no longer tell the layers apart; the dispatch order does (rows appear in it). Repeated runs with the same key and a
# duration within 1.5 % of an earlier row's are folded into that row (a bench loop's rounds).
import os
FOLD = float(os.environ.get("PMC_FOLD", os.environ.get("PMC_TOL", "0.015")))   # PMC_FOLD=0: every run its own row (the probe runs each layer once)
TOL = float(os.environ.get("PMC_TOL", "0.015"))   # PMC_TOL=0.25 for the layer probe: its runs are separated by other kernels already, and a counter pass is noisy
last = OrderedDict()
prev_key, run, run_ns = None, 0, 0.0
for d in disp.values():
    key = (d["name"], d["grid"])
    if key != prev_key or abs(d["ns"] - run_ns) > TOL * run_ns:    # (conv2 and conv3 follow each other on one kernel and one grid)
        run += 1
        prev_key, run_ns = key, d["ns"]
    last[(d["name"], d["grid"], run)] = d
seen, folded = [], OrderedDict()
for (n, g, r), d in last.items():
    if any(n == n0 and g == g0 and abs(d["ns"] - ns0) <= FOLD * ns0 for n0, g0, ns0 in seen):
        continue
    seen.append((n, g, d["ns"]))
    folded[(n, "%s#%d" % (g, r))] = d
last = folded
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return re.sub(r"\(.*$", "", n)[:44]
print("%-44s %13s %9s %6s %6s %6s %6s %6s %8s %6s" % ("kernel", "grid", "vgpr+a", "mfma", "w_any", "w_inst", "active", "ldsbc", "us", "GHz"))
for (n, g), d in last.items():
    if "wgrad" not in n and "conv" not in n: continue
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    gui = d.get("GRBM_GUI_ACTIVE", 0) or 1
    print("%-44s %13s %9s %6.3f %6.2f %6.2f %6.2f %6.3f %8.1f %6.2f" % (short(n), g, "%s+%s" % (d["vgpr"], d["agpr"]),
          d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 128), d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc,
          d.get("SQ_ACTIVE_INST_ANY", 0) / wc, d.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, d.get("SQ_ACTIVE_INST_LDS", 0)),
          d["ns"] / 1e3, (gui / d["ns"]) if d["ns"] else 0.0))   # effective shader clock = GRBM_GUI_ACTIVE / wall (profiled pass)
