#!/bin/bash
# kernel start / end times of the overlap experiment (rocprofv3 --kernel-trace): do the two streams' kernels run side by side at all?
export TMPDIR=/tmp
O=gpurun_out/r04_ovl; mkdir -p $O; rm -rf $O/trace
export QM2_LIB=$PWD/variant_ovl.so QM2_EXP_T_PRIO=1 QM2_EXP_T_GRID=${1:-64}
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-streamed > $O/trace_bench.json 2> $O/trace.err
python3 - $O/trace <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
keep = [r for r in rows if any(s in r["Kernel_Name"] for s in ("split_text", "split_kmer", "bucket_count"))]
t0 = int(keep[0]["Start_Timestamp"])
for r in keep[:40]:
    print("%-14s queue %s  start %9.3f ms  end %9.3f ms  grid %s" % (r["Kernel_Name"].split("<")[0].split("::")[-1][:14], r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - t0) / 1e6, r.get("Grid_Size_X", r.get("Grid_Size"))))
PY
