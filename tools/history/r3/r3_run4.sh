#!/bin/bash
O=gpurun_out/r3d
mkdir -p $O
QM2_E2E_REPS=2 QM2_E2E_RUNS="fa:12,fa:12:s0,fq:12:s0,fq:12:0::host" timeout -k 10 400 python tools/e2e_cli.py 40000000 /dev/shm/qm2_e2e c2 > $O/cli_c2_shards.txt 2>&1; echo "c2 rc=$?"; grep "^==" $O/cli_c2_shards.txt; tail -22 $O/cli_c2_shards.txt
QM2_SHARDS_FAKE_EXCHANGE=1 QM2_E2E_REPS=1 QM2_E2E_RUNS="fa:12:s0+0,fa:12:s0+0+0+0" timeout -k 10 400 python tools/e2e_cli.py 40000000 /dev/shm/qm2_e2e c2 > $O/cli_c2_shards_fake.txt 2>&1; echo "c2 fake rc=$?"; grep "^==" $O/cli_c2_shards_fake.txt; tail -14 $O/cli_c2_shards_fake.txt
