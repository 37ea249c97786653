#!/bin/bash
O=gpurun_out/r3x
mkdir -p $O
QM2_FORCE_DIST=1 QM2_SHARDED_JOB=1 timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --parallelism sharded --reads-per-step 20000000 > $O/bench_sharded_job_1rank_rehearsal.json 2> $O/shjob.err; echo "sharded job rehearsal rc=$?"; tail -3 $O/shjob.err
python -c "
import json; d=json.load(open('$O/bench_sharded_job_1rank_rehearsal.json')); print(round(d['value']/1e9,2), d['config'].get('counts_conserved'), d.get('streamed_bases_per_s'))"
