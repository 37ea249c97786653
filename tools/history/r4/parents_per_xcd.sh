#!/bin/bash
# EXPERIMENT: split_kmer's workgroups of one XCD work on F level-1 parents at a time instead of one (-DQM2_EXP_PARENTS_PER_XCD=F)
export QM2_TAG=r04_ppx
echo "product (F = 1):"; bash tools/gpu_run.sh quick || exit 1
for f in 2 4 16; do echo "F = $f:"; bash tools/gpu_run.sh variant variant_ppx$f.so ppx$f || exit 1; done
echo "product (F = 1):"; bash tools/gpu_run.sh quick
