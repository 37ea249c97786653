#!/bin/bash
# round 3: after the prepare-thread fix (the thread makes the partition buffers only): CLI tests, CLI timing at C3 scale
O=gpurun_out/r3i
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cli or devices or shards or hint" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
QM2_E2E_REPS=2 QM2_E2E_RUNS="fq:16,fq:16:0+0" timeout -k 10 700 python tools/e2e_cli.py 100000000 /tmp/qm2_e2e c3 /dev/shm/qm2_reads > $O/cli_c3_100M.txt 2>&1; echo "e2e c3 rc=$?"; grep "^==\|QM2 phase\|QM2 epilogue" $O/cli_c3_100M.txt | head -30
