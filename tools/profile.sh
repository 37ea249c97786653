#!/bin/bash
# tools/profile.sh <tag> [bench args...] — run on the GPU box (via gpurun).
# Writes rocprofv3 output under gpurun_out/prof_<tag>/ :
#   trace/   --kernel-trace --stats of bench.py (per-kernel durations)
#   pmc_rd/  --pmc FETCH_SIZE          (separate pass, as MI355X_MICROARCH.md prescribes)
#   pmc_wr/  --pmc WRITE_SIZE
#   pmc_sq/  --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ...  (the instruction side: roofline.valu)
#   pmc_ea/  --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum  (request counts)
# plus the same PMC passes over tools/microbench.py (known access counts: calibration).
set -u
TAG=${1:-r01}; shift || true
ROOTDIR=$(pwd)
OUT=$ROOTDIR/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-streamed --no-geometry-check --no-random-access --no-realistic $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || echo "trace pass failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_rd" -- python3 bench.py $ARGS > "$OUT/bench_pmc_rd.json" 2> "$OUT/pmc_rd.err" || echo "pmc_rd failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_wr" -- python3 bench.py $ARGS > "$OUT/bench_pmc_wr.json" 2> "$OUT/pmc_wr.err" || echo "pmc_wr failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py $ARGS > "$OUT/bench_pmc_sq.json" 2> "$OUT/pmc_sq.err" || echo "pmc_sq failed"
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_ATOMIC_sum --output-format csv -d "$OUT/pmc_ea" -- python3 bench.py $ARGS > "$OUT/bench_pmc_ea.json" 2> "$OUT/pmc_ea.err" || echo "pmc_ea failed"
if [ "${QM2_PROFILE_CAL:-0}" = "1" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/cal_rd" -- python3 tools/microbench.py 28 > "$OUT/cal_rd.jsonl" 2> "$OUT/cal_rd.err" || echo "cal_rd failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/cal_wr" -- python3 tools/microbench.py 28 > "$OUT/cal_wr.jsonl" 2> "$OUT/cal_wr.err" || echo "cal_wr failed"
fi
python3 tools/summarize_profile.py "$OUT" "${QM2_WORKLOAD:-c3}" "profiles/${TAG}" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep only small files for the merge back
find "$OUT" -name "*.db" -delete 2>/dev/null
du -sh "$OUT"
