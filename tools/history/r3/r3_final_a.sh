#!/bin/bash
# round 3, final evidence of the final tree, part A: rocprofv3 kernel stats + PMC traffic + SQ counters, the default bench line
O=gpurun_out/r3x
mkdir -p $O
bash tools/profile.sh r03_c3 > $O/profile.log 2>&1; tail -3 $O/profile.log
cp gpurun_out/traffic_c3.json profiles/traffic_c3.json
bash tools/pmc_sq.sh r03_c3 > $O/sq_counters.txt 2> $O/pmc_sq.err
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(round(d['value']/1e9,2), d['streamed_bases_per_s'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'] if d['cpu_baseline'] else None, d['config']['partitioned_equals_direct_at_bench_batch'])"
