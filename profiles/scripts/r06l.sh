#!/bin/bash
# round 6: the in-loop flux form, the remaining cases -- fp32 at P <= 5 (shipped PREG build against geom:50), P = 9, 10 (prev library against the
# tree's), and P = 7 round by round (its median was 8 % slower than its minimum in r06k)
O=gpurun_out/r06l
mkdir -p $O
for cfg in "2 107" "3 71" "4 54" "5 43"; do
  set -- $cfg
  timeout -k 10 300 python tools/ab_stiffness.py --degree $1 --cells $2 --dtype f32 --rounds 5 --reps 100 geom geom:50 2>&1 | grep -v "Warning\|amdgpu.ids"
done | tee $O/ab_geom_flux_form_f32.log
for cfg in "9 24 f64" "10 21 f64" "7 31 f64" "7 31 f32" "6 36 f32" "8 27 f32"; do
  set -- $cfg
  for rep in 1 2; do
    for lib in prev tree; do
      if [ $lib = tree ]; then l=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else l=$PWD/tools/_bin/libfusgpu_prev.so; fi
      FUS_LIB_PATH=$l timeout -k 10 300 python tools/ab_stiffness.py --degree $1 --cells $2 --dtype $3 --rounds 9 --reps 100 geom 2>&1 | grep "^geom\|rounds:" | sed "s/^/P=$1 $3 $lib: /"
    done
  done
done | tee $O/ab_flux_form_rest.log
