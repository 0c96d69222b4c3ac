#!/bin/bash
# round 6: the in-loop flux form (column_flux_at) at P >= 6 -- the library BEFORE it (tools/_bin/libfusgpu_prev.so: a build of the previous commit's
# stiffness_geom.hpp / westervelt_geom.hpp) against the tree's, alternating processes on one box, 100-launch bursts; stiffness and the Westervelt step
O=gpurun_out/r06k
mkdir -p $O
for cfg in "6 36" "7 31" "8 27"; do
  set -- $cfg
  for rep in 1 2; do
    for lib in prev tree; do
      if [ $lib = tree ]; then l=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else l=$PWD/tools/_bin/libfusgpu_prev.so; fi
      FUS_LIB_PATH=$l timeout -k 10 300 python tools/ab_stiffness.py --degree $1 --cells $2 --rounds 5 --reps 100 geom 2>&1 | grep "^geom" | sed "s/^/P=$1 $lib: /"
    done
  done
done | tee $O/ab_flux_form_p678.log
for rep in 1 2; do
  for lib in prev tree; do
    if [ $lib = tree ]; then l=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else l=$PWD/tools/_bin/libfusgpu_prev.so; fi
    FUS_LIB_PATH=$l python bench.py --mode westervelt --degree 6 --cells 36 --in-kernel-geometry --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; o=json.loads(sys.stdin.read()); print(sys.argv[1], "westervelt geom step ms %.4f" % o["ms_per_step"], "check %.2e" % o["check"]["rel_l2"])' $lib
  done
done | tee $O/ab_flux_form_westervelt.log
