#!/bin/bash
# what the last gpurun call of round 4 ran: the suite, the driver's bench line, the N > 1 code path on one rank over RCCL
export QM2_TAG=r04
O=gpurun_out/r04; mkdir -p $O
bash tools/gpu_run.sh suite || exit 1
bash tools/gpu_run.sh bench || exit 1
QM2_FORCE_DIST=1 timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_replica_1rank_dist_rehearsal.json 2> $O/bench_replica_1rank_dist_rehearsal.err || { echo "dist rehearsal failed"; tail -5 $O/bench_replica_1rank_dist_rehearsal.err; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04/bench_replica_1rank_dist_rehearsal.json") if l.startswith("{")][-1])
print("1-rank RCCL rehearsal:", round(d["value"] / 1e9, 1), "G bases/s, reduce", d["config"].get("final_reduce_ms"), "ms, conserved", d["config"].get("counts_conserved"), "streamed per rank", (d.get("streamed") or {}).get("h2d_GBps_per_rank"))
PY
