#!/bin/bash
O=gpurun_out/r3z
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "merge or outgrow" > $O/t3.log 2>&1; echo "rc=$?"; tail -8 $O/t3.log | cut -c1-220
