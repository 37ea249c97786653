#!/bin/bash
# EXPERIMENT: streaming (nontemporal) loads for what is read once — split_kmer's source k-mers (-DQM2_NT_KMER_FETCH=1),
# bucket_count's records (-DQM2_NT_RECORDS=1) and table image (-DQM2_NT_IMAGE=1) — so that they do not push the open write fronts
# (split_kmer) and the counter lines (bucket_count) out of L2
export QM2_TAG=r04_nt
echo "product:"; bash tools/gpu_run.sh quick || exit 1
for v in ntk ntr nti ntall; do echo "$v:"; bash tools/gpu_run.sh variant variant_$v.so $v || exit 1; done
echo "product:"; bash tools/gpu_run.sh quick
