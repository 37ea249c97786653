#!/bin/bash
# round 3, GPU call 1: MALL ring micro-benchmark, the new geometry parity test, the bench with its geometry check,
# and two timing-only variants of bucket_count (edge atomics off / half the counter traffic)
O=gpurun_out/r3a
mkdir -p $O
timeout -k 10 240 ./tools/micro/ring_probe 300 > $O/ring_probe.txt 2>&1; echo "ring_probe rc=$?"; tail -5 $O/ring_probe.txt
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cli_batch_geometry" > $O/geom_test.log 2>&1; echo "geom test rc=$?"; tail -3 $O/geom_test.log
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for f in ["bench"]:
    try:
        d=json.load(open("gpurun_out/r3a/%s.json"%f)); print(f, round(d['value']/1e9,2), 'G bases/s', {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()}, d['config'].get('partitioned_equals_direct_at_bench_batch'), d.get('streamed_bases_per_s'))
    except Exception as e: print(f, "failed", e)
PY
for v in NO_EDGE_ATOMICS NO_EDGE_HALF_COUNTERS; do
  QM2_LIB=$PWD/quick-mer2_amd/variants_$v.so timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed --no-geometry-check > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v rc=$?"
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', round(d['value']/1e9,2), 'G bases/s', {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()})"
done
