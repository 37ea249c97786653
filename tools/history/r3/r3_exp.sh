#!/bin/bash
O=gpurun_out/r3x
mkdir -p $O
run() { n=$1; shift; "$@" > $O/$n.json 2> $O/$n.err || { echo "$n FAILED"; tail -3 $O/$n.err; return; }
  python -c "
import json; d=json.load(open('$O/$n.json')); print('%-16s' % '$n', round(d['value']/1e9,2), 'G bases/s', {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()}, d['config']['partitioned_equals_direct_at_bench_batch'])"; }
B="timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed"
run base60 $B
run cpw1 env QM2_SPLIT_CHUNKS_PER_WG=1 $B
run cpw2 env QM2_SPLIT_CHUNKS_PER_WG=2 $B
run cpw3 env QM2_SPLIT_CHUNKS_PER_WG=3 $B
run cpw2_r64 env QM2_SPLIT_CHUNKS_PER_WG=2 $B --reads-per-step 64000000
run cpw2_r50 env QM2_SPLIT_CHUNKS_PER_WG=2 $B --reads-per-step 50000000
run cpw2_c2 env QM2_SPLIT_CHUNKS_PER_WG=2 $B --workload c2
run base_c2 $B --workload c2
