#!/bin/bash
# round 3 evidence, part 2: whole GPU suite + smoke, the N>1 bench code path rehearsed on one rank over RCCL, the CLI on a 300 M-read FASTQ
O=gpurun_out/r3g
mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
QM2_FORCE_DIST=1 timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $O/bench_replica_1rank_dist_rehearsal.json 2> $O/dist.err; echo "dist rehearsal rc=$?"
QM2_FORCE_DIST=1 QM2_SHARDED_JOB=1 timeout -k 10 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --parallelism sharded --reads-per-step 20000000 > $O/bench_sharded_job_1rank_rehearsal.json 2> $O/shjob.err; echo "sharded job rehearsal rc=$?"
python - <<'PY'
import json
for f in ["bench_replica_1rank_dist_rehearsal","bench_sharded_job_1rank_rehearsal"]:
    try:
        d=json.load(open("gpurun_out/r3g/%s.json"%f)); print(f, round(d['value']/1e9,2), d.get('streamed_bases_per_s'), (d.get('streamed') or {}).get('h2d_GBps_per_rank'), d['config'].get('final_reduce_ms'))
    except Exception as e: print(f, "failed", e)
PY
QM2_E2E_REPS=2 QM2_E2E_RUNS="fq:16" timeout -k 10 1000 python tools/e2e_cli.py 300000000 /tmp/qm2_e2e c3 /dev/shm/qm2_reads > $O/cli_c3_300M.txt 2>&1; echo "e2e 300M rc=$?"; grep "^==\|QM2 phase\|QM2 epilogue" $O/cli_c3_300M.txt | tail -8
