#!/bin/bash
# part C: the CLI end to end: hg38 scale 100 M and 300 M reads (FASTQ), chr22 scale FASTA / FASTQ / two contexts
O=gpurun_out/r3x
mkdir -p $O
QM2_E2E_REPS=2 QM2_E2E_RUNS="fq:12,fq:16" timeout -k 10 500 python tools/e2e_cli.py 100000000 /tmp/qm2_e2e c3 /dev/shm/qm2_reads > $O/cli_c3_100M.txt 2>&1; echo "e2e 100M rc=$?"; grep "^==" $O/cli_c3_100M.txt | cut -c1-330
rm -rf /dev/shm/qm2_reads /tmp/qm2_e2e
QM2_E2E_REPS=2 QM2_E2E_RUNS="fq:16" timeout -k 10 600 python tools/e2e_cli.py 300000000 /tmp/qm2_e2e c3 /dev/shm/qm2_reads > $O/cli_c3_300M.txt 2>&1; echo "e2e 300M rc=$?"; grep "^==\|QM2 phase\|QM2 epilogue" $O/cli_c3_300M.txt | tail -8 | cut -c1-330
