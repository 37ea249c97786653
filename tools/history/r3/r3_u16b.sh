#!/bin/bash
O=gpurun_out/r3z
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "outgrow or wrap or deep or shards or merge" > $O/t2.log 2>&1; echo "rc=$?"; tail -12 $O/t2.log | cut -c1-220
