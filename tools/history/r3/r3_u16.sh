#!/bin/bash
O=gpurun_out/r3z
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -15 $O/gpu_tests.log | cut -c1-200
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-streamed > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench.json')); print('product', round(d['value']/1e9,2), 'G bases/s', {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()}, d['config']['partitioned_equals_direct_at_bench_batch'], d['config'].get('counts_conserved'))"
