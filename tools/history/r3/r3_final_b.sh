#!/bin/bash
# part B: the other workloads, the whole GPU suite, smoke(), the N > 1 code path rehearsed on one rank over RCCL
O=gpurun_out/r3x
mkdir -p $O
timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "c5 rc=$?"
timeout -k 10 300 python bench.py --workload c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
python -c "
import json
for w in ('c5','c2'):
    d=json.load(open('$O/bench_%s.json' % w)); print(w, round(d['value']/1e9,2), d['roofline']['frac'], d['config'].get('partitioned_equals_direct_at_bench_batch'))"
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
QM2_FORCE_DIST=1 timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_replica_1rank_dist_rehearsal.json 2> $O/dist.err; echo "dist rehearsal rc=$?"
