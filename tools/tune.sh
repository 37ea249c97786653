#!/bin/bash
# tools/tune.sh — bench every variants_*.so (built with different -D tuning flags) on one box
for so in quick-mer2_amd/libqm2_hip.so variants_*.so; do
  [ -f "$so" ] || continue
  echo "== $so"
  QM2_LIB=$PWD/$so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e9,2),'Gb/s', d['ms_per_step'], {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()})"
done
