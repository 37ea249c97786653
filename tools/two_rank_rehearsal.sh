#!/bin/bash
# tools/two_rank_rehearsal.sh [n] — bench.py's N = 2 replica path on ONE GPU (gloo collectives), n times; prints what each run found
export QM2_REHEARSE_ON_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/${QM2_TAG:-r04}_two; mkdir -p $O
for i in $(seq 1 ${1:-4}); do
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29620 + i)) bench.py --gpus 2 --steps 3 --warmup 1 --workload c2 --reads-per-step 4000000 --mode ${QM2_MODE:-partition} --no-streamed --no-cpu-baseline --no-geometry-check > $O/run$i.json 2> $O/run$i.err || { echo "run $i failed"; tail -5 $O/run$i.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('$O/run$i.json') if l.startswith('{')][-1]); print('run $i', round(d['value']/1e9,1), 'conserved', d['config']['counts_conserved'])"
  grep "NOT CONSERVED" $O/run$i.err
done
exit 0
