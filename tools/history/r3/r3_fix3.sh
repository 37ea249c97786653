#!/bin/bash
# round 3: C2-scale CLI, FASTA and FASTQ, after the reader change
O=gpurun_out/r3m
mkdir -p $O
QM2_E2E_REPS=3 QM2_E2E_RUNS="fq:8,fq:12,fq:16,fa:8,fa:12,fa:16,fa:12:0+0" timeout -k 10 500 python tools/e2e_cli.py 40000000 /dev/shm/qm2_e2e c2 > $O/cli_c2_40M.txt 2>&1; echo "e2e c2 rc=$?"; grep "^==" $O/cli_c2_40M.txt
