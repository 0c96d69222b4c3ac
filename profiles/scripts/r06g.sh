#!/bin/bash
# round 6, VERDICT r5 item 6: the low-degree mass apply -- float-atomic batch plan | transposed-dofmap gather | its static-detJ form,
# interleaved (tools/ab_mass_gather.py), P = 2 and 3, fp64 and fp32, ~10 M dofs; rows per thread 0 (auto), 1, 2, 4
O=gpurun_out/r06g
mkdir -p $O
for cfg in "2 107 f64" "2 107 f32" "3 71 f64" "3 71 f32" "4 54 f32"; do
  set -- $cfg
  timeout -k 10 300 python tools/ab_mass_gather.py --degree $1 --cells $2 --dtype $3 --rounds 5 --reps 30 --variants 0,1,2,4 2>&1 | grep -v Warning
done | tee $O/ab_mass_low_degree.log
