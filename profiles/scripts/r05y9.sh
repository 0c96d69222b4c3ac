#!/bin/bash
# round 5: the every-thread preamble extended to P = 7, 8 (scratch build, plan_loads_by_all: n <= 9) vs the shipped switch at P <= 6
O=gpurun_out/r05y
mkdir -p $O
for v in tree n9 tree n9; do
  if [ $v = tree ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_n9.so; fi
  echo "== $v"
  FUS_LIB_PATH=$lib timeout -k 10 400 python tools/sweep.py --degrees 7,8 2>&1 | grep "^P=" | sed 's/| M .*//'
done | tee $O/sweep_ab_p78.log
