#!/bin/bash
# A/B of the vector half of the fused RK4 step: LEAN stage kinds 4-7 (csrc/rk4.hpp) against kinds 2, 0, 0, 3,
# interleaved, linear step (config 3, in-kernel geometry = the solver's default) and Westervelt step (config 5 shape).
#   bash tools/ab_rk4_lean.sh > profiles/r06e_ab_rk4_lean.log
show='import json,sys; o=json.loads(sys.stdin.read()); print(sys.argv[1], "ms/step %.4f" % o["ms_per_step"], "touches", o["roofline"]["vector_touches_per_step"], "frac %.3f" % o["roofline"]["frac"], "check rel_l2 %.2e" % o["check"]["rel_l2"], o["check"]["ok"])'
for rep in 1 2 3; do
  for lean in 1 0; do
    FUS_RK4_LEAN=$lean python bench.py --mode rk4 --perturbed --in-kernel-geometry --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$show" "linear geom  lean=$lean"
  done
done
for rep in 1 2; do
  for lean in 1 0; do
    FUS_RK4_LEAN=$lean python bench.py --mode rk4 --perturbed --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$show" "linear G     lean=$lean"
  done
done
for rep in 1 2 3; do
  for lean in 1 0; do
    FUS_RK4_LEAN=$lean python bench.py --mode westervelt --degree 6 --cells 36 --in-kernel-geometry --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$show" "westervelt   lean=$lean"
  done
done
