#!/bin/bash
# EXPERIMENT (r4): split_text and bucket_count keep one workgroup resident per CU, so their grid IS the number of CUs they use:
# QM2_EXP_NCU sizes the grids for fewer CUs (the others idle).  Which of the two scales with its CUs, which with the memory system?
# (A first attempt with hipExtStreamCreateWithCUMask gave steps of whole rounds — 256 resident workgroups on fewer CUs — and is
# of no use for these two; split_kmer, whose workgroups come and go, scaled linearly with the CUs of the mask.)
export QM2_TAG=r04_cus QM2_EXP_NO_OVL=1
for n in 256 224 192 160 128 96 64; do
  QM2_EXP_NCU=$n bash tools/gpu_run.sh variant variant_ovl.so ncu$n || exit 1
done
