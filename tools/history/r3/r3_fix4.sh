#!/bin/bash
# round 3: where the epilogue's worker threads spend their time at hg38 scale
O=gpurun_out/r3n
mkdir -p $O
QM2_E2E_REPS=2 QM2_E2E_RUNS="fq:12" timeout -k 10 500 python tools/e2e_cli.py 20000000 /tmp/qm2_e2e c3 /dev/shm/qm2_reads > $O/cli_c3_20M.txt 2>&1; echo "e2e c3 rc=$?"; grep "^==\|QM2 epilogue" $O/cli_c3_20M.txt
