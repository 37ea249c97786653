#!/bin/bash
# usage: tools/sweep_args.sh "args1" "args2" ...   — one bench process per argument string, per-kernel ms printed
for a in "$@"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-streamed $a 2>/tmp/sweep.err > /tmp/sweep.json || { echo "$a failed"; continue; }
  python - "$a" <<'PY'
import sys, json
d = json.loads(open('/tmp/sweep.json').read())
print(sys.argv[1], "|", round(d["value"] / 1e9, 2), {k: round(v["ms_per_step"], 2) for k, v in d["roofline"]["kernels"].items() if v["ms_per_step"] > 1})
PY
  grep "STAMPS split" /tmp/sweep.err | head -3
done
