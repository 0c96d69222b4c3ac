#!/bin/bash
# round 6 experiment: wave priority in the in-kernel-geometry kernel (tools/exp_geom_prio.py: scratch builds), alternating processes, P = 4, 6
O=gpurun_out/r06p
mkdir -p $O
for cfg in "4 54" "6 36"; do
  set -- $cfg
  for rep in 1 2; do
    for lib in tree prio prio2; do
      if [ $lib = tree ]; then l=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else l=$PWD/tools/_bin/libfusgpu_$lib.so; fi
      FUS_LIB_PATH=$l timeout -k 10 300 python tools/ab_stiffness.py --degree $1 --cells $2 --rounds 7 --reps 100 geom 2>&1 | grep "^geom" | sed "s/^/P=$1 $lib: /"
    done
  done
done | tee $O/ab_geom_prio.log
