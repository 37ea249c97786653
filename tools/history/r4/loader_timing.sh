#!/bin/bash
# EXPERIMENTS on top of the loaders' hold-back: (a) the probers' record prefetch in front of the first probe set again
# (QM2_PREFETCH_LATE=0, variant_early.so) with the hold-back at 8 / 12 / 16; (b) the hold-back split in two: h steps at the head of the
# period, m steps between flush and stage (variant_mid.so, QM2_EXP_MID_SLEEP)
export QM2_TAG=r04_lt
echo "product:"; bash tools/gpu_run.sh quick || exit 1
for hs in 8 12 16; do echo "early prefetch, head sleep $hs:"; QM2_COUNT_HEAD_SLEEP=$hs bash tools/gpu_run.sh variant variant_early.so early_hs$hs || exit 1; done
for hm in "0 12" "4 8" "8 4" "0 8" "6 8" "12 4"; do set -- $hm; echo "head $1 + mid $2:"; QM2_COUNT_HEAD_SLEEP=$1 QM2_EXP_MID_SLEEP=$2 bash tools/gpu_run.sh variant variant_mid.so h$1_m$2 || exit 1; done
echo "product:"; bash tools/gpu_run.sh quick
