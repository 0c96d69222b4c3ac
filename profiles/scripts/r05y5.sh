#!/bin/bash
# round 5: the geometry factor's division as v_rcp_f64 + two Newton steps (scratch build tools/_bin/libfusgpu_rcp.so) vs the IEEE division:
# steady-state time (tools/ablate_geom.py) and the bench line's check against the oracle, in-kernel geometry mode
O=gpurun_out/r05y
mkdir -p $O
{
for v in tree rcp tree rcp; do
  if [ $v = tree ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_rcp.so; fi
  FUS_LIB_PATH=$lib timeout -k 10 200 python tools/ablate_geom.py 2>&1 | grep -E "^P=" | sed "s/^/$v: /"
done
for v in tree rcp; do
  if [ $v = tree ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_rcp.so; fi
  FUS_LIB_PATH=$lib timeout -k 10 300 python bench.py --mode stiffness_geom --no-aux --no-cpu-baseline > $O/bench_geom_$v.json 2> $O/bench_geom_$v.err || { tail -5 $O/bench_geom_$v.err; exit 1; }
  python -c "
import json,sys
d=json.loads(open('$O/bench_geom_$v.json').read().strip().splitlines()[-1])
print('$v: bench --mode stiffness_geom', round(d['ms_per_step'],4), 'ms; check', d['check']['rel_l2'], d['check']['rel_max'])"
done
} | tee $O/ab_rcp_division.log
