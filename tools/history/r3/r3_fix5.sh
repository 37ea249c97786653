#!/bin/bash
# round 3: GC sums on the device: API test, CLI tests (every .txt of the reference), epilogue timing at hg38 scale
O=gpurun_out/r3o
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "gc_sums or cli or chain or epilogue" > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
QM2_E2E_REPS=2 QM2_E2E_RUNS="fq:12" timeout -k 10 500 python tools/e2e_cli.py 20000000 /tmp/qm2_e2e c3 /dev/shm/qm2_reads > $O/cli_c3_20M.txt 2>&1; echo "e2e c3 rc=$?"; grep "^==\|QM2 epilogue" $O/cli_c3_20M.txt
