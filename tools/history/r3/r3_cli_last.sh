#!/bin/bash
O=gpurun_out/r3last
mkdir -p $O
QM2_E2E_REPS=2 QM2_E2E_RUNS="fq:12,fq:12:0::host" timeout -k 10 500 python tools/e2e_cli.py 100000000 /tmp/qm2_e2e c3 /dev/shm/qm2_reads > $O/cli_c3_100M.txt 2>&1; echo "e2e 100M rc=$?"; grep "^==\|QM2 phase\|QM2 epilogue" $O/cli_c3_100M.txt | cut -c1-330 | tail -8
