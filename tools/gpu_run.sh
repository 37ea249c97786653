#!/bin/bash
# tools/gpu_run.sh <recipe> [args...] — the recipes a GPU-box call (gpurun) of this repo runs, one at a time or chained:
#   gpurun --timeout 900 -- 'bash tools/gpu_run.sh suite && bash tools/gpu_run.sh bench'
# Everything is written under gpurun_out/<tag>/ (tag = $QM2_TAG, default r04); what is to be kept goes to profiles/ by hand.
# Every GPU step runs under its own `timeout -k 10`; a step that times out stops the recipe (no further GPU step).
set -u
TAG=${QM2_TAG:-r05}
O=gpurun_out/$TAG
mkdir -p "$O"
export TMPDIR=/tmp

step() {                      # step <seconds> <log> <command...>: run, report, stop the recipe on a timeout / failure
    local secs=$1 log=$2; shift 2
    timeout -k 10 "$secs" "$@" > "$log" 2> "${log%.*}.err"
    local rc=$?
    echo "[$(date +%H:%M:%S)] rc=$rc  $*  -> $log"
    [ $rc -eq 0 ] || { tail -5 "${log%.*}.err"; tail -5 "$log"; exit 1; }
}

line() {                      # the figures of a bench line that one looks at first
    python3 - "$1" <<'EOF'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("%.2f G bases/s, %.3f ms/step, frac %.4f | %s | host-fed packed %s, FASTQ %s (ASCII %s, raw FASTQ %s) | realistic %s | exact at the bench batch: %s | conserved: %s" % (
    d["value"] / 1e9, d["ms_per_step"], r["frac"],
    {k: round(v["ms_per_step"], 2) for k, v in r.get("kernels", {}).items() if v["ms_per_step"] > 0.05} or r.get("kernel_ms_per_step"),
    d.get("streamed_bases_per_s"), (d.get("streamed") or {}).get("streamed_fastq_bases_per_s"), (d.get("streamed") or {}).get("streamed_ascii_bases_per_s"),
    (d.get("streamed") or {}).get("streamed_raw_fastq_bases_per_s"), d.get("realistic_bases_per_s"),
    d["config"].get("partitioned_equals_direct_at_bench_batch"), d["config"].get("counts_conserved")))
EOF
}

case "${1:-help}" in
suite)        # the whole GPU suite + smoke (what the driver runs at round end)
    step 1100 $O/gpu_tests.log python3 -m pytest tests -m gpu -x -q
    tail -2 $O/gpu_tests.log
    step 200 $O/smoke.log python3 -c "import __graft_entry__ as g; g.smoke()"
    tail -1 $O/smoke.log ;;
tests)        # a subset: tools/gpu_run.sh tests "<pytest -k expression>"
    step 900 $O/gpu_tests_k.log python3 -m pytest tests -m gpu -x -q -k "$2"
    tail -3 $O/gpu_tests_k.log ;;
bench)        # the driver's own command line (default workload), full line
    shift
    step 500 $O/bench_default.json python3 bench.py --steps 20 --warmup 5 "$@"
    line $O/bench_default.json ;;
quick)        # kernel work: a short bench without the CPU legs and streamed regions; extra args go to bench.py
    shift
    step 300 $O/bench_quick${QM2_SUFFIX:-}.json python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed --no-random-access --no-realistic "$@"
    line $O/bench_quick${QM2_SUFFIX:-}.json ;;
variant)      # the same with another build of the library: tools/gpu_run.sh variant <lib.so> <suffix> [bench args]
    lib=$2; suf=$3; shift 3
    QM2_LIB=$PWD/$lib step 300 $O/bench_variant_$suf.json python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed --no-random-access --no-realistic "$@"
    line $O/bench_variant_$suf.json
    grep STAMPS $O/bench_variant_$suf.err || true ;;
profile)      # rocprofv3 kernel trace + the PMC passes (tools/profile.sh), the SQ counters; summary under gpurun_out/prof_<tag>_<wl>/
    wl=${2:-c3}
    QM2_WORKLOAD=$wl bash tools/profile.sh ${TAG}_$wl --workload $wl > $O/profile_$wl.log 2>&1; tail -3 $O/profile_$wl.log
    bash tools/pmc_sq.sh ${TAG}_$wl > $O/sq_counters_$wl.txt 2> $O/pmc_sq_$wl.err || true ;;
others)       # the other workloads' lines
    step 400 $O/bench_c5.json python3 bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline
    line $O/bench_c5.json
    step 300 $O/bench_c2.json python3 bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline
    line $O/bench_c2.json ;;
cli)          # the CLI end to end: tools/gpu_run.sh cli <reads> <c2|c3>
    step 1000 $O/cli_${3:-c3}_${2:-100000000}.txt python3 tools/e2e_cli.py ${2:-100000000} /tmp/qm2_e2e ${3:-c3} /dev/shm/qm2_e2e
    tail -30 $O/cli_${3:-c3}_${2:-100000000}.txt ;;
help|*)
    sed -n 2,8p "$0"; grep -E '^[a-z]+\)' "$0" ;;
esac
