#!/bin/bash
# EXPERIMENT: the three prober waves of a SIMD start a period 0 / n / 2n x 64 cycles apart (variant_ps.so, -DQM2_EXP_PROBER_STAGGER)
export QM2_TAG=r04_ps
for n in 0 2 4 8 16; do
  echo "prober stagger $n:"
  QM2_EXP_PROBER_STAGGER_N=$n bash tools/gpu_run.sh variant variant_ps.so ps$n || exit 1
done
