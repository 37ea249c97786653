#!/bin/bash
# bench the product build and the variant libs named on the command line (no tests): one line each
O=gpurun_out/r3v_$1; shift
mkdir -p $O
run() { n=$1; shift; "$@" timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed $BENCH_ARGS > $O/bench_$n.json 2> $O/bench_$n.err || { echo "$n FAILED"; tail -3 $O/bench_$n.err; return; }
  python -c "
import json; d=json.load(open('$O/bench_$n.json')); print('%-14s' % '$n', round(d['value']/1e9,2), 'G bases/s', {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()}, d['config']['partitioned_equals_direct_at_bench_batch'])"; }
run product env
for v in "$@"; do run $v env QM2_LIB=$PWD/quick-mer2_amd/variants_$v.so; done
