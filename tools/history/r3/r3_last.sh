#!/bin/bash
O=gpurun_out/r3last
mkdir -p $O
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(round(d['value']/1e9,2), d['roofline']['frac'], d['config']['partitioned_equals_direct_at_bench_batch'], d['streamed']['streamed_bases_per_s'])"
