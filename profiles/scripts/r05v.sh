#!/bin/bash
# round 5: mass_gather_kernel with every load issued by every lane (clamped indices, batch loads pinned) vs the committed form: degree sweep,
# mass columns (gather default / static detJ / float-atomic), fp64 and fp32, interleaved twice
O=gpurun_out/r05v
mkdir -p $O
for v in prev tree prev tree; do
  if [ $v = prev ]; then lib=$PWD/tools/_bin/libfusgpu_prev.so; else lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; fi
  echo "== $v"
  FUS_LIB_PATH=$lib timeout -k 10 400 python tools/sweep.py --degrees 2,3,4,6,8 2>&1 | grep "^P=" | sed 's/K\[plan\].*| M /M /'
done | tee $O/sweep_mass_ab.log
