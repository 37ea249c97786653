#!/bin/bash
# Address-translation counters of split_text in several processes (its time differs between processes:
# EXPERIMENTS.md section 4.1): one rocprofv3 --pmc pass per process, the bench's own kernel times beside the counters.
set -u
OUT=$(pwd)/gpurun_out/pmc_tlb_${1:-a}
N=${2:-5}
mkdir -p $OUT
export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-streamed"
for i in $(seq 1 $N); do
  timeout -k 10 240 rocprofv3 --pmc ${PMC:-TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY} --output-format csv -d $OUT/p$i -- python3 bench.py $A > $OUT/b$i.json 2> $OUT/p$i.err || { echo "pass $i failed"; grep -m2 "error code\|Could not" $OUT/p$i.err; exit 1; }
  python3 - "$OUT" $i <<'PY'
import csv, glob, json, os, sys
from collections import defaultdict
out, i = sys.argv[1], sys.argv[2]
try:
    d = json.load(open(os.path.join(out, "b%s.json" % i)))
    print("pass", i, {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items() if v["ms_per_step"] > 1})
except Exception as e:
    print("pass", i, "bench line unreadable:", e)
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p" + i, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "split" in k or "bucket_count" in k:
        print("   ", k[:34], {c.replace("TCP_UTCL1_", "").replace("_sum", ""): "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
  find $OUT -name "*.db" -delete
done
