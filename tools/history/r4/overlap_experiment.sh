#!/bin/bash
# EXPERIMENT (r4): the text split of batch n + 1 on a second stream beside bucket_count of batch n (variant_ovl.so built with
# -DQM2_EXP_OVERLAP), its grid limited to QM2_EXP_T_GRID workgroups so that bucket_count keeps the other CUs;
# QM2_EXP_T_PRIO=1: that stream with the highest priority (without it the text split's workgroups wait until bucket_count's are all placed)
export QM2_TAG=r04_ovl
bash tools/gpu_run.sh quick || exit 1
for tg in 0 160 128 96 64 32; do
  QM2_EXP_T_PRIO=1 QM2_EXP_T_GRID=$tg bash tools/gpu_run.sh variant variant_ovl.so prio_tg$tg || exit 1
done
bash tools/gpu_run.sh quick
