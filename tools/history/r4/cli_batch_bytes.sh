#!/bin/bash
# the CLI at hg38 scale (100 M-read FASTQ) with smaller device batches: does the load phase shrink with the partition scratch,
# and does the count phase stay at the PCIe rate?
O=gpurun_out/r04_bb; mkdir -p $O
for bb in 0 3221225472 2147483648; do
  if [ $bb = 0 ]; then unset QM2_BATCH_BYTES; else export QM2_BATCH_BYTES=$bb; fi
  QM2_E2E_RUNS="fq:12" QM2_E2E_REPS=5 timeout -k 10 600 python3 tools/e2e_cli.py 100000000 /tmp/qm2_e2e c3 /dev/shm/qm2_e2e > $O/bb_$bb.txt 2> $O/bb_$bb.err || { echo "failed at $bb"; tail -5 $O/bb_$bb.err; exit 1; }
  echo "batch bytes $bb:"; grep "^== \|rep [0-9]" $O/bb_$bb.txt | cut -c1-330
done
