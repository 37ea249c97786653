#!/bin/bash
# EXPERIMENT: split_kmer's workgroups per parent from the expected fill (x QM2_EXP_WPP_FILL) instead of the parent's capacity
export QM2_TAG=r04_wpp
echo "product:"; bash tools/gpu_run.sh quick || exit 1
for f in 1.0 1.1 0.75 0.5; do echo "fill x $f:"; QM2_EXP_WPP_FILL=$f QM2_SUFFIX=_wpp$f bash tools/gpu_run.sh quick || exit 1; done
for f in 1.0; do for w in 1 3; do echo "fill x $f, chunks per workgroup $w:"; QM2_SPLIT_CHUNKS_PER_WG=$w QM2_EXP_WPP_FILL=$f QM2_SUFFIX=_wpp${f}_$w bash tools/gpu_run.sh quick || exit 1; done; done
