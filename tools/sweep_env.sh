#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   — one bench process per value, per-kernel ms printed
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed 2>/dev/null > /tmp/sweep.json || { echo "$var=$v failed"; continue; }
  python - "$var=$v" <<'PY'
import sys, json
d = json.loads(open('/tmp/sweep.json').read())
print(sys.argv[1], round(d["value"] / 1e9, 2), {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items() if v["ms_per_step"] > 1})
PY
done
