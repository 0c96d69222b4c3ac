#!/bin/bash
# round 5: run tables vs raw dof lists with the restructured preamble (the run words are now read speculatively, before nu[batch] has
# arrived): is the auto choice (fp64: run tables at every degree; fp32: up to P = 4) still right?
O=gpurun_out/r05x
mkdir -p $O
{
for spec in "f32 5 43 1" "f32 6 36 1" "f32 7 31 1" "f32 8 27 1" "f32 4 54 30" "f32 2 107 30" "f64 2 107 0" "f64 4 54 1" "f64 6 36 2" "f64 8 27 2"; do
  set -- $spec
  echo "== $1 P=$2 cells=$3^3 build $4"
  timeout -k 10 300 python tools/ab_stiffness.py --dtype $1 --degree $2 --cells $3 --rounds 5 --reps 20 plan raw:$4 runs:$4 2>&1 | grep -v "^\[" | tail -4
done
} | tee $O/ab_run_tables.log
