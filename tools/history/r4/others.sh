export QM2_TAG=r04
O=gpurun_out/r04; mkdir -p $O
bash tools/gpu_run.sh others || exit 1
timeout -k 10 500 python3 tools/heavy_hitters.py 50000000 > $O/heavy_hitters.txt 2> $O/heavy_hitters.err || { echo "heavy hitters failed"; tail -5 $O/heavy_hitters.err; exit 1; }
tail -12 $O/heavy_hitters.txt
timeout -k 10 400 python3 bench.py --parallelism sharded --shards 8 --reads-per-step 20000000 --steps 6 --warmup 2 --no-cpu-baseline --no-streamed > $O/bench_sharded_8_logical_20M.json 2> $O/bench_sharded_8.err || { echo "sharded failed"; tail -5 $O/bench_sharded_8.err; exit 1; }
timeout -k 10 400 python3 bench.py --reads-per-step 20000000 --steps 6 --warmup 2 --no-cpu-baseline --no-streamed --no-geometry-check > $O/bench_replica_20M.json 2> $O/bench_replica_20M.err || exit 1
python3 - <<'PY'
import json
for f in ("bench_sharded_8_logical_20M", "bench_replica_20M"):
    d = json.loads([l for l in open("gpurun_out/r04/%s.json" % f) if l.startswith("{")][-1])
    print(f, round(d["value"] / 1e9, 1), "G bases/s", d["config"].get("counts_conserved"), d["roofline"].get("kernel_ms_per_step") or {k: round(v["ms_per_step"], 2) for k, v in d["roofline"].get("kernels", {}).items()})
PY
