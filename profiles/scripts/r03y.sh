#!/bin/bash
set -e
O=gpurun_out/r03y
mkdir -p $O
FUS_BENCH_REHEARSAL=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/torchrun2.json 2> $O/torchrun2.err || { tail -30 $O/torchrun2.err; exit 1; }
python -c "
import json; d=json.loads([l for l in open('$O/torchrun2.json') if l.startswith('{')][-1]); c=d['config']
print(d['n_gpus'], round(d['ms_per_step'],3), c['halo_transport'][:30], c['halo_check']['ok'], c['halo_schedule'], c['halo_transports_tried'])"
