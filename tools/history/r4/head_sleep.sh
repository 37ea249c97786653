#!/bin/bash
# bucket_count: the loader waves hold back QM2_COUNT_HEAD_SLEEP x 256 cycles at the head of every period — by batch size
export QM2_TAG=r04_hs
for r in ${QM2_HS_READS:-60000000 40000000 20000000}; do
  for hs in ${QM2_HS_VALUES:-0 4 6 8 10 12 16}; do
    echo "reads per step $r, head sleep $hs:"
    QM2_COUNT_HEAD_SLEEP=$hs QM2_SUFFIX=_${r}_hs$hs bash tools/gpu_run.sh quick --no-geometry-check --reads-per-step $r || exit 1
  done
done
