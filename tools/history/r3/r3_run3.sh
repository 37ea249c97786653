#!/bin/bash
# round 3: the CLI end to end at hg38 scale (100 M reads = 30.7 GB of FASTQ in /dev/shm), thread counts, and chr22 scale with 40 M reads for --devices
O=gpurun_out/r3c
mkdir -p $O
df -h /dev/shm /tmp | tail -2
QM2_E2E_REPS=2 QM2_E2E_RUNS="fq:12,fq:16,fq:24,fq:32" timeout -k 10 900 python tools/e2e_cli.py 100000000 /tmp/qm2_e2e c3 /dev/shm/qm2_reads > $O/cli_c3_100M.txt 2>&1; echo "e2e c3 rc=$?"; grep "^==\|QM2 phase\|QM2 epilogue" $O/cli_c3_100M.txt | head -30
QM2_E2E_REPS=3 QM2_E2E_RUNS="fq:12,fq:16,fq:12:0+0,fq:16:0+0,fq:12:0+0:host,fa:12,fa:12:0+0" timeout -k 10 400 python tools/e2e_cli.py 40000000 /dev/shm/qm2_e2e c2 > $O/cli_c2_40M.txt 2>&1; echo "e2e c2 rc=$?"; grep "^==" $O/cli_c2_40M.txt
