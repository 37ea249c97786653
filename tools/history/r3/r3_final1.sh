#!/bin/bash
# round 3 evidence, part 1: rocprofv3 kernel stats + PMC traffic + SQ counters of the final build, the default bench line
O=gpurun_out/r3f
mkdir -p $O
bash tools/profile.sh r03_c3 > $O/profile.log 2>&1; tail -5 $O/profile.log
bash tools/pmc_sq.sh r03_c3 > $O/sq_counters.txt 2> $O/pmc_sq.err; tail -3 $O/sq_counters.txt | cut -c1-200
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(round(d['value']/1e9,2), d['streamed_bases_per_s'], d['roofline']['frac'], d['cpu_baseline']['value'] if d['cpu_baseline'] else None, d['config']['partitioned_equals_direct_at_bench_batch'])"
