#!/bin/bash
# round 3: pread from the page cache into different kinds of destination memory
O=gpurun_out/r3k
mkdir -p $O
nproc > $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>/dev/null
timeout -k 10 400 tools/micro/pread_probe /dev/shm/qm2_pread_probe.bin 8 2 > $O/pread_probe.txt 2>&1; echo "rc=$?"; cat $O/nproc.txt; cat $O/pread_probe.txt
rm -f /dev/shm/qm2_pread_probe.bin
