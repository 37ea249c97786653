#!/bin/bash
# round 3: after the SSE2 newline count in the reader: CLI tests, CLI timing at C3 scale by thread count, then C2 FASTA/FASTQ
O=gpurun_out/r3l
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cli or devices or line_discipline" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -eq 0 ] || exit 1
QM2_E2E_REPS=2 QM2_E2E_RUNS="fq:8,fq:12,fq:16,fq:24,fq:16:0+0" timeout -k 10 800 python tools/e2e_cli.py 100000000 /tmp/qm2_e2e c3 /dev/shm/qm2_reads > $O/cli_c3_100M.txt 2>&1; echo "e2e c3 rc=$?"; grep "^==" $O/cli_c3_100M.txt
