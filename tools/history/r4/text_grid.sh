#!/bin/bash
# EXPERIMENT: split_text with workgroups that come and go (grid = CUs x QM2_EXP_T_GRID_MULT) instead of one resident workgroup per CU
export QM2_TAG=r04_tg
for m in 1 2 4 16 64 256 1024; do
  echo "grid mult $m:"; QM2_EXP_T_GRID_MULT=$m QM2_SUFFIX=_m$m bash tools/gpu_run.sh quick || exit 1
done
