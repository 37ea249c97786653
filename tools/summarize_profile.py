#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a summary: per-kernel
stats (calls, total/avg ns) and per-kernel PMC sums (per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                yield f, r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
seen = False
for f, r in rows("trace/**/*kernel_stats.csv"):
    seen = True
    print("%-70s calls=%s total_ns=%s avg_ns=%s pct=%s" % (r.get("Name", "")[:70], r.get("Calls"), r.get("TotalDurationNs"),
                                                           r.get("AverageNs"), r.get("Percentage")))
if not seen:
    # fall back to the trace itself
    agg = defaultdict(lambda: [0, 0])
    for f, r in rows("trace/**/*kernel_trace.csv"):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[r["Kernel_Name"]][0] += 1
        agg[r["Kernel_Name"]][1] += d
    for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
        print("%-70s calls=%d total_ns=%d avg_ns=%.0f" % (k[:70], c, t, t / c))

for sub in ("pmc_rd", "pmc_wr", "pmc_ea", "cal_rd", "cal_wr"):
    agg = defaultdict(lambda: defaultdict(list))
    for f, r in rows(sub + "/**/*counter_collection.csv"):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not agg:
        continue
    print("== %s: per-kernel counter values (n dispatches, mean per dispatch, last) ==" % sub)
    for k, cs in agg.items():
        for cn, vals in cs.items():
            print("%-60s %-22s n=%d mean=%.6g last=%.6g" % (k[:60], cn, len(vals), sum(vals) / len(vals), vals[-1]))
