#!/bin/bash
# round 5 experiment: ~128-thread workgroups (P = 4: 5 cells per batch; scratch build) vs the shipped ~256-thread ones at 1 M and 10 M dofs
O=gpurun_out/r05y
mkdir -p $O
for cells in 25 32 54; do
  for v in tree cpb128 tree cpb128; do
    if [ $v = tree ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_cpb128.so; fi
    FUS_LIB_PATH=$lib timeout -k 10 300 python tools/ab_stiffness.py --cells $cells --rounds 5 --reps 50 plan geom 2>&1 | grep -E "^plan|^geom" | sed "s/^/cells=$cells $v: /"
  done
done | tee $O/ab_cpb128.log
