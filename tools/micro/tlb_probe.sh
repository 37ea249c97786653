#!/bin/bash
# tools/micro/tlb_probe.sh [out-dir] — tlb_probe under rocprofv3 --pmc, its lines joined with the UTCL1 counters per dispatch
set -u
O=${1:-$(pwd)/gpurun_out/tlbp}
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_UTCL2_BUSY --output-format csv -d $O/p -- ./tools/micro/tlb_probe > $O/run.txt 2> $O/err.txt || { echo "tlb_probe failed"; tail -3 $O/err.txt; exit 1; }
python3 - $O <<'PY'
import csv, glob, os, sys
from collections import defaultdict
rows = defaultdict(dict)
for f in glob.glob(os.path.join(sys.argv[1], "p", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "streams_kernel" in r["Kernel_Name"]:
            rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
lines = [l.rstrip() for l in open(os.path.join(sys.argv[1], "run.txt")) if l.startswith("dispatch")]
for l, k in zip(lines, sorted(rows)):
    c = rows[k]
    print("%s | UTCL1 requests %.3g misses %.3g, UTCL2 busy cycles %.3g" % (l, c.get("TCP_UTCL1_REQUEST_sum", 0), c.get("TCP_UTCL1_TRANSLATION_MISS_sum", 0),
                                                                             c.get("GRBM_UTCL2_BUSY", 0)))
PY
find $O -name "*.db" -delete
