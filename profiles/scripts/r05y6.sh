#!/bin/bash
# ablation: the general-G kernel (and every planned kernel) WITHOUT its flush (no global atomics), vs the shipped library: what the y update costs
for v in tree noflush tree noflush; do
  if [ $v = tree ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_noflush.so; fi
  FUS_LIB_PATH=$lib timeout -k 10 300 python tools/ab_stiffness.py --rounds 5 --reps 20 plan geom 2>&1 | grep -E "^plan|^geom" | sed "s/^/$v: /"
done
