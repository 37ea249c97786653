#!/bin/bash
O=gpurun_out/r3s
mkdir -p $O
run() { n=$1; shift; timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed --no-geometry-check --reads-per-step 20000000 "$@" > $O/$n.json 2> $O/$n.err || { echo "$n FAILED"; tail -5 $O/$n.err; return; }
  python -c "
import json; d=json.load(open('$O/$n.json')); r=d['roofline']; print('%-28s' % '$n', round(d['value']/1e9,2), 'G bases/s', r.get('kernel_ms_per_step') or {k: v['ms_per_step'] for k, v in r['kernels'].items()}, d['config'].get('counts_conserved'))"; }
run bench_replica_20M
run bench_sharded_8_logical_20M --parallelism sharded --shards 8
run bench_sharded_8_logical_20M_flat --parallelism sharded --shards 8 --flat-slabs
run bench_sharded_2_logical_20M --parallelism sharded --shards 2
