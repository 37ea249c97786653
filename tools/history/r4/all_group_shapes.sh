#!/bin/bash
# every instantiation of bucket_count_kernel (rows per group 2..6 x loader waves 3 / 4, each with 2- and 4-byte second words
# through the tests' table sizes) forced through the partition-sensitive GPU tests: the launch normally picks one per batch
O=gpurun_out/r04_shapes; mkdir -p $O
for g in 2 3 4 5 6; do for lw in 3 4; do
  QM2_COUNT_G=$g QM2_COUNT_LW=$lw timeout -k 10 400 python3 -m pytest tests -m gpu -x -q -k "c3_scale_partitioned or c2_scale or api_mini or skew or long_occ or other_k or kat_cases or c4" > $O/g${g}_lw$lw.log 2>&1
  echo "G=$g LW=$lw rc=$? $(tail -1 $O/g${g}_lw$lw.log)"
done; done
