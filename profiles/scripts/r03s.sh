#!/bin/bash
set -e
O=gpurun_out/r03s
mkdir -p $O
FUS_BENCH_REHEARSAL=1 timeout -k 10 500 python bench.py --gpus 4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rehearsal4.json 2> $O/bench_rehearsal4.err || { tail -20 $O/bench_rehearsal4.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench_rehearsal4.json')); c=d['config']
print(d['n_gpus'], d['ms_per_step'], c['halo_transport'][:40], c['halo_check'], c['halo_schedule'], c['partition'], c['global_dofs'])"
FUS_BENCH_REHEARSAL=1 timeout -k 10 500 python bench.py --gpus 2 --mode rk4 --perturbed --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_rehearsal2_rk4.json 2> $O/bench_rehearsal2_rk4.err || { tail -20 $O/bench_rehearsal2_rk4.err; exit 1; }
head -c 400 $O/bench_rehearsal2_rk4.json; echo
