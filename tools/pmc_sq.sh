#!/bin/bash
# SQ-level counters for the partitioned pipeline's kernels (two passes of 8 counters)
set -u
OUT=$(pwd)/gpurun_out/pmc_sq_${1:-a}
mkdir -p $OUT
export TMPDIR=/tmp
A="--steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-geometry-check --no-random-access --no-realistic"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p1 -- python3 bench.py $A > $OUT/b1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- python3 bench.py $A > $OUT/b2.json 2> $OUT/p2.err
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for sub in ("p1", "p2"):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if "split" in k or "bucket_count" in k or "count_kernel" in k:
            print(k[:50], {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()})
PY
find $OUT -name "*.db" -delete
