#!/bin/bash
# round 3, GPU call 2: whole GPU suite; CLI end to end at chr22 scale (threads, --devices 0,0 raw vs host); sharded bench on one stream
O=gpurun_out/r3b
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $O/gpu_tests.log
QM2_E2E_REPS=3 QM2_E2E_RUNS="fq:8,fq:12,fq:16,fq:24,fq:32,fa:12,fa:24,fq:12:0+0,fq:24:0+0,fq:12:0+0:host,fq:12:0,fa:12:0+0" timeout -k 10 400 python tools/e2e_cli.py 10000000 /dev/shm/qm2_e2e c2 > $O/cli_c2.txt 2>&1; echo "e2e c2 rc=$?"; grep "^==" $O/cli_c2.txt
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed --no-geometry-check --reads-per-step 20000000 > $O/bench_replica_20M.json 2> $O/bench_replica_20M.err; echo "replica 20M rc=$?"
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --parallelism sharded --shards 8 --reads-per-step 20000000 > $O/bench_sharded_8_logical_20M.json 2> $O/bench_sharded_8.err; echo "sharded rc=$?"
python - <<'PY'
import json
for f in ["bench_replica_20M","bench_sharded_8_logical_20M"]:
    try:
        d=json.load(open("gpurun_out/r3b/%s.json"%f)); r=d['roofline']
        print(f, round(d['value']/1e9,2), 'G bases/s', r.get('kernel_ms_per_step') or {k: v['ms_per_step'] for k, v in r['kernels'].items()}, d['config'].get('spilled_kmers', d['config'].get('spilled_kmers_rank0_or_all_logical')))
    except Exception as e: print(f, "failed", e)
PY
