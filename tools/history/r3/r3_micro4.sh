#!/bin/bash
O=gpurun_out/r3p
mkdir -p $O
timeout -k 10 300 tools/micro/superkmer_probe 20000000 > $O/superkmer_probe.txt 2>&1; echo "rc=$?"; cat $O/superkmer_probe.txt
