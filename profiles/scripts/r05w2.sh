#!/bin/bash
# the 6-process mass rehearsal three times (time-sliced processes on one GPU: the figure is NOT a measurement, only its stability is looked at)
export FUS_BENCH_REHEARSAL=1 FUS_IPC_SPIN_SECONDS=60
mkdir -p gpurun_out/r05w
for i in 1 2 3; do
  timeout -k 10 500 python bench.py --gpus 6 --mode mass --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05w/mass_$i.json 2> gpurun_out/r05w/mass_$i.err || { tail -20 gpurun_out/r05w/mass_$i.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r05w/mass_$i.json'))
print('run $i', round(d['ms_per_step'],3), d['config'].get('halo_check',{}).get('ok'), (d.get('check') or {}).get('rel_l2'))"
done
