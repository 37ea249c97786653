#!/bin/bash
# round 3, final evidence at the final default (60 M reads per step): rocprofv3 kernel stats + PMC traffic + SQ counters, then the default
# bench line with the traffic of THIS build, the other workloads, the whole GPU suite and smoke()
O=gpurun_out/r3h
mkdir -p $O
bash tools/profile.sh r03_c3 > $O/profile.log 2>&1; tail -3 $O/profile.log
cp gpurun_out/traffic_c3.json profiles/traffic_c3.json
bash tools/pmc_sq.sh r03_c3 > $O/sq_counters.txt 2> $O/pmc_sq.err
timeout -k 10 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"
python -c "
import json; d=json.load(open('$O/bench_default.json')); print(round(d['value']/1e9,2), d['streamed_bases_per_s'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'] if d['cpu_baseline'] else None, d['config']['partitioned_equals_direct_at_bench_batch'])"
timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
python -c "
import json
for w in ('c5','c2'):
    d=json.load(open('$O/bench_%s.json' % w)); print(w, round(d['value']/1e9,2), d['roofline']['frac'], d['config'].get('partitioned_equals_direct_at_bench_batch'))"
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
