#!/bin/bash
# round 3: the copy-out the other way round (a lane per run) against the rows of today, stores only
O=gpurun_out/r3j
mkdir -p $O
timeout -k 10 300 tools/micro/front_probe 2 1400 > $O/front_probe.txt 2>&1; echo "rc=$?"; head -40 $O/front_probe.txt
