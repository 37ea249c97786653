#!/bin/bash
# round 3 kernel iteration: partition-sensitive parity subset + C3 bench (geometry check on) [+ variant libs given as arguments]
TAG=${1:-q}; shift
O=gpurun_out/r3q_$TAG
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c3_scale or c2_scale or api_mini or batch_bound or skewed or long_occ or other_k or fuzz or route or shards_with or geometry or c5_hifi or c4_eight" > $O/tests.log 2>&1
rc=$?; tail -3 $O/tests.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-streamed > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json')); print('product', round(d['value']/1e9,2), 'G bases/s', {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()}, d['config']['partitioned_equals_direct_at_bench_batch'])"
for v in "$@"; do
  QM2_LIB=$PWD/quick-mer2_amd/variants_$v.so timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed > $O/bench_$v.json 2> $O/bench_$v.err || { tail -3 $O/bench_$v.err; continue; }
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', round(d['value']/1e9,2), 'G bases/s', {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()}, d['config']['partitioned_equals_direct_at_bench_batch'])"
done
