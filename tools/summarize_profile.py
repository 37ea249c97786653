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

# ---- traffic_<workload>.json: HBM bytes per launch and per step from the PMC passes ------------------
# FETCH_SIZE / WRITE_SIZE are KiB per dispatch, collected in SEPARATE passes; on gfx950 FETCH_SIZE tallies
# a wide coalesced read at half its bytes (MI355X_MICROARCH.md, HBM), so the read side is doubled.
import json
import re

if len(sys.argv) > 2:
    workload, dest = sys.argv[2], sys.argv[3]
    pk = defaultdict(dict)
    for sub, cn in (("pmc_rd", "FETCH_SIZE"), ("pmc_wr", "WRITE_SIZE")):
        agg = defaultdict(list)
        for f, r in rows(sub + "/**/*counter_collection.csv"):
            if r["Counter_Name"] == cn:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for kname, vals in agg.items():
            m = re.search(r"qm2::(\w+)_kernel", kname)
            if not m or m.group(1) not in ("split_text", "split_kmer", "bucket_count", "count", "probe"):
                continue
            cls = {"count": "count_direct"}.get(m.group(1), m.group(1))
            e = pk[cls]
            e.setdefault(cn + "_KiB", 0.0)
            e.setdefault(cn + "_n", 0)
            e[cn + "_KiB"] += sum(vals)
            e[cn + "_n"] += len(vals)
    # the instruction side (tools/profile.sh, pass pmc_sq): wave-instructions and LDS cycles per dispatch
    for cn, key in (("SQ_INSTS_VALU", "valu_wave_insts_per_launch"), ("SQ_INSTS_LDS", "lds_insts_per_launch"),
                    ("SQ_LDS_BANK_CONFLICT", "lds_bank_conflict_cycles_per_launch"), ("SQ_LDS_IDX_ACTIVE", "lds_active_cycles_per_launch")):
        agg = defaultdict(list)
        for f, r in rows("pmc_sq/**/*counter_collection.csv"):
            if r["Counter_Name"] == cn:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        for kname, vals in agg.items():
            m = re.search(r"qm2::(\w+)_kernel", kname)
            if not m or m.group(1) not in ("split_text", "split_kmer", "bucket_count", "count", "probe"):
                continue
            cls = {"count": "count_direct"}.get(m.group(1), m.group(1))
            pk[cls].setdefault("_sq", {})
            e = pk[cls]["_sq"].setdefault(key, [0.0, 0])
            e[0] += sum(vals)
            e[1] += len(vals)
    steps = None
    try:
        bj = json.load(open(os.path.join(out, "bench_pmc_rd.json")))
        steps = bj["steps"] + bj["warmup"]
    except Exception:
        pass
    per_kernel, step_total = {}, 0.0
    for cls, e in pk.items():
        n = max(e.get("FETCH_SIZE_n", 0), e.get("WRITE_SIZE_n", 0), 1)
        rd = e.get("FETCH_SIZE_KiB", 0.0) / max(1, e.get("FETCH_SIZE_n", 1))
        wr = e.get("WRITE_SIZE_KiB", 0.0) / max(1, e.get("WRITE_SIZE_n", 1))
        hbm = int(2 * rd * 1024 + wr * 1024)
        per_kernel[cls] = {"FETCH_SIZE_KiB": round(rd), "WRITE_SIZE_KiB": round(wr), "hbm_bytes_per_launch": hbm,
                           "launches_profiled": n}
        for key, (tot, cnt) in e.get("_sq", {}).items():
            per_kernel[cls][key] = round(tot / max(1, cnt))
        if steps:
            step_total += hbm * n / steps
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench as _bench
    tj = {"workload": workload,
          "kernel_source_sha16": _bench.kernel_source_sha16(),      # bench.py quotes these counters only for the kernels they were measured on
          "source": "%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py` (tools/profile.sh), "
                    "condensed by tools/summarize_profile.py; KiB per dispatch; the read side is doubled (gfx950 tallies wide "
                    "coalesced reads at half their bytes: MI355X_MICROARCH.md, HBM); SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_LDS_* from a "
                    "third pass, per dispatch" % os.path.relpath(dest, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")),
          "per_kernel": per_kernel,
          "hbm_bytes_per_step": int(step_total) if steps else None}
    json.dump(tj, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gpurun_out", "traffic_%s.json" % workload), "w"), indent=1)
    print("== traffic_%s.json ==" % workload)
    print(json.dumps(tj, indent=1))
