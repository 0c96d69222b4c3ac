#!/bin/bash
# round 6: the flux without forming G (column_flux_at: 55 instead of 67 fp64 operations per quadrature point) -- (a) P <= 5: shipped build (the n x 6
# factors in registers, formed between the gather's barriers) against the in-loop operator form at 4 waves per SIMD (geom:50, 12 B of scratch at P = 4)
# and at the compiler's own allocation (geom:51: 3 waves), interleaved, 100-launch bursts; (b) P >= 6 and the Westervelt cell pass take it
# unconditionally: their lines before / after are the sweep and the bench line
O=gpurun_out/r06j
mkdir -p $O
for cfg in "4 54" "3 71" "5 43" "2 107"; do
  set -- $cfg
  timeout -k 10 300 python tools/ab_stiffness.py --degree $1 --cells $2 --rounds 5 --reps 100 geom geom:50 geom:51 2>&1 | grep -v Warning
done | tee $O/ab_geom_flux_form.log
( timeout -k 10 600 python tools/sweep.py --degrees 6,7,8 --dtypes f64 2>&1 | grep "^P=" ) | tee $O/sweep_p678.log | cut -c1-200
FUS_RK4_LEAN=1 python bench.py --mode westervelt --degree 6 --cells 36 --in-kernel-geometry --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; o=json.loads(sys.stdin.read()); print("westervelt geom step ms", o["ms_per_step"], "check", o["check"]["rel_l2"], o["check"]["ok"])' | tee $O/westervelt_step.log
