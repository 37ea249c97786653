#!/bin/bash
O=gpurun_out/r3w
mkdir -p $O
run() { n=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed --no-geometry-check > $O/$n.json 2> $O/$n.err || { echo "$n FAILED"; tail -3 $O/$n.err; return; }
  python -c "
import json; d=json.load(open('$O/$n.json')); print('%-12s' % '$n', round(d['value']/1e9,2), {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()}, d['config'].get('counts_conserved'))"; }
run default A=1
run lw3 QM2_COUNT_LW=3
run lw3g5 QM2_COUNT_LW=3 QM2_COUNT_G=5
run lw4g6 QM2_COUNT_LW=4 QM2_COUNT_G=6
run lw3g6 QM2_COUNT_LW=3 QM2_COUNT_G=6
run lw4g4 QM2_COUNT_LW=4 QM2_COUNT_G=4
