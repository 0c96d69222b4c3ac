#!/bin/bash
# round 5: run tables vs raw lists for fp32 at P = 7, 8 after the every-thread preamble reached those degrees
O=gpurun_out/r05x
mkdir -p $O
{
for spec in "f32 7 31 1" "f32 8 27 1" "f32 6 36 1"; do
  set -- $spec
  echo "== $1 P=$2 cells=$3^3 build $4"
  timeout -k 10 300 python tools/ab_stiffness.py --dtype $1 --degree $2 --cells $3 --rounds 5 --reps 20 plan raw:$4 runs:$4 2>&1 | grep -v "^\[" | tail -3
done
} | tee $O/ab_run_tables_fp32_p78.log
