#!/bin/bash
# the bench default at other batch sizes (reads per step), final tree
export QM2_TAG=r04_sizes
for r in 10000000 20000000 30000000 40000000 50000000 60000000 70000000; do
  QM2_SUFFIX=_$r bash tools/gpu_run.sh quick --no-geometry-check --reads-per-step $r || exit 1
done
