#!/bin/bash
O=gpurun_out/r3s8
mkdir -p $O
run() { n=$1; lib=$2; shift 2; env $lib timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed --no-geometry-check "$@" > $O/$n.json 2> $O/$n.err || { echo "$n FAILED"; tail -3 $O/$n.err; return; }
  python -c "
import json; d=json.load(open('$O/$n.json')); print('%-16s' % '$n', round(d['value']/1e9,2), {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()}, d['config'].get('counts_conserved'))"; }
V=QM2_LIB=$PWD/quick-mer2_amd/variants_set8.so
for w in "c2:--workload c2" "c5:--workload c5" "c3_40M:--reads-per-step 40000000" "c3_20M:--reads-per-step 20000000" "c3_50M:--reads-per-step 50000000"; do n=${w%%:*}; a=${w#*:}; run ${n}_set6 A=1 $a; run ${n}_set8 $V $a; done
