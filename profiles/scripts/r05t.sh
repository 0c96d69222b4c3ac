#!/bin/bash
# round 5: degree sweep (tools/sweep.py) of two builds of the restructured preamble: tools/_bin/libfusgpu_prev.so (loads of the preamble
# under ``active``) and the library of this tree (loads by every thread; slots narrowed after the gather is issued)
O=gpurun_out/r05t
mkdir -p $O
for v in prev tree prev tree; do
  if [ $v = prev ]; then lib=$PWD/tools/_bin/libfusgpu_prev.so; else lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; fi
  echo "== $v"
  FUS_LIB_PATH=$lib timeout -k 10 400 python tools/sweep.py --degrees 2,3,4,5,6,7,8 2>&1 | grep "^P=" | sed 's/| M .*//'
done | tee $O/sweep_ab.log
