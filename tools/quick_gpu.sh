#!/bin/bash
# tools/quick_gpu.sh — one GPU-box round for kernel work: a parity subset, the C3 bench, and the phase
# stamps of the diagnostic build (variants_stamps.so) when present.  Output under gpurun_out/<tag>_*.
TAG=${1:-q}
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c3_scale or c2_scale or api_mini or batch_bound or skewed or long_occ or other_k or fuzz" > gpurun_out/${TAG}_tests.log 2>&1
rc=$?
tail -3 gpurun_out/${TAG}_tests.log
[ $rc -eq 124 ] && exit 1
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || exit 1
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print(round(d['value']/1e9,2), 'G bases/s', {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()})"
if [ -f variants_stamps.so ]; then
  QM2_LIB=$PWD/variants_stamps.so timeout -k 10 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_stamps.json 2> gpurun_out/${TAG}_stamps.err
  grep STAMPS gpurun_out/${TAG}_stamps.err
fi
