export QM2_REHEARSE_ON_ONE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/r04_stress; mkdir -p $O
for i in 1 2 3 4 5 6; do
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29700 + i)) bench.py --gpus 2 --steps 40 --warmup 2 --workload c2 --reads-per-step 1500000 --mode partition --no-streamed --no-cpu-baseline --no-geometry-check > $O/run$i.json 2> $O/run$i.err || { echo "run $i failed"; tail -5 $O/run$i.err; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/run$i.json') if l.startswith('{')][-1]); print('run $i', round(d['value']/1e9,1), 'conserved', d['config']['counts_conserved'])"
  grep "NOT CONSERVED" $O/run$i.err | cut -c1-300
done
for i in 7 8 9; do
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port $((29700 + i)) bench.py --gpus 3 --steps 12 --warmup 2 --workload c2 --reads-per-step 4000000 --mode partition --no-streamed --no-cpu-baseline --no-geometry-check > $O/run$i.json 2> $O/run$i.err || { echo "run $i failed"; tail -5 $O/run$i.err; exit 1; }
  python3 -c "
import json
d=json.loads([l for l in open('$O/run$i.json') if l.startswith('{')][-1]); print('run $i (3 ranks)', round(d['value']/1e9,1), 'conserved', d['config']['counts_conserved'])"
  grep "NOT CONSERVED" $O/run$i.err | cut -c1-300
done
exit 0
