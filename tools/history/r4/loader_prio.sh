#!/bin/bash
# EXPERIMENT: the loader waves' s_setprio (2 in the product) with and without the head-of-period hold-back
export QM2_TAG=r04_prio
for p in 0 3; do for hs in 0 6 12; do
  echo "loader prio $p, head sleep $hs:"
  QM2_COUNT_HEAD_SLEEP=$hs bash tools/gpu_run.sh variant variant_prio$p.so p${p}_hs$hs || exit 1
done; done
echo "product (prio 2, rule):"; bash tools/gpu_run.sh quick
