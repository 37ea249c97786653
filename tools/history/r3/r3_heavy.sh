#!/bin/bash
O=gpurun_out/r3y
mkdir -p $O
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streamed --no-geometry-check --reads-per-step 50000000 > $O/bench50.json 2> $O/bench50.err
python -c "
import json; d=json.load(open('$O/bench50.json')); print('bench 50M', round(d['value']/1e9,2), {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()})"
HH_ONLY=uniform HH_KPK=0 QM2_PLAN_PRINT=1 timeout -k 10 400 python tools/heavy_hitters.py 50000000 > $O/hh_nohint.txt 2> $O/hh_nohint.err; cat $O/hh_nohint.txt | cut -c1-260; grep "QM2 plan" $O/hh_nohint.err | tail -2
QM2_PLAN_PRINT=1 timeout -k 10 900 python tools/heavy_hitters.py 50000000 > $O/heavy_hitters.txt 2> $O/heavy.err; echo "rc=$?"; tail -12 $O/heavy_hitters.txt | cut -c1-260; grep "QM2 plan" $O/heavy.err | tail -2
