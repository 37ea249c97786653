#!/bin/bash
# three-stage prober variants: parity subset with the variant library, then benches
O=gpurun_out/r3p_$1; shift
mkdir -p $O
V=$1
QM2_LIB=$PWD/quick-mer2_amd/variants_$V.so timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c3_scale or c2_scale or api_mini or batch_bound or skewed or long_occ or other_k or fuzz or route or shards_with or geometry or c5_hifi or c4_eight or wrap_mod or empty_and" > $O/tests_$V.log 2>&1
rc=$?; tail -3 $O/tests_$V.log
[ $rc -ne 0 ] && { grep -n "Error\|assert" $O/tests_$V.log | head -20; }
run() { n=$1; shift; "$@" timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed $BENCH_ARGS > $O/bench_$n.json 2> $O/bench_$n.err || { echo "$n FAILED"; tail -3 $O/bench_$n.err; return; }
  python -c "
import json; d=json.load(open('$O/bench_$n.json')); print('%-14s' % '$n', round(d['value']/1e9,2), 'G bases/s', {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()}, d['config']['partitioned_equals_direct_at_bench_batch'])"; }
run product env
for v in "$@"; do run $v env QM2_LIB=$PWD/quick-mer2_amd/variants_$v.so; done
