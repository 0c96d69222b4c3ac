#!/bin/bash
# round 5 experiment: exclusive dofs of the general-G kernel finished with a plain load + store (scratch build) vs the shipped atomics
O=gpurun_out/r05y
mkdir -p $O
for v in tree excl tree excl; do
  if [ $v = tree ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_excl.so; fi
  FUS_LIB_PATH=$lib timeout -k 10 300 python tools/exp_excl_flush.py 2>&1 | grep -v "amdgpu.ids\|^\[" | sed "s/^/$v: /"
done | tee $O/exp_excl_flush.log
