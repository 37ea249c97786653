#!/bin/bash
# usage: tools/sweep.sh "bench args" VAR v1 v2 ...  — one bench process per value of VAR
args=$1; var=$2; shift 2
for v in "$@"; do
  env $var=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-streamed $args 2>/tmp/sweep.err > /tmp/sweep.json || { echo "$var=$v failed"; tail -3 /tmp/sweep.err; continue; }
  python - "$var=$v" <<'PY'
import sys, json
d = json.loads(open('/tmp/sweep.json').read())
print(sys.argv[1], "|", round(d["value"] / 1e9, 2), {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items() if v["ms_per_step"] > 1})
PY
done
