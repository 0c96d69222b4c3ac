#!/bin/bash
# round 5: ablation of the in-kernel-geometry kernel (builds of a COPY of csrc/ with -DFUS_ABLATE bits; the shipped sources are untouched)
O=gpurun_out/r05o
mkdir -p $O
for rep in 1 2; do
  for a in "" 1 2 3 4 7; do
    if [ -z "$a" ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_ablate$a.so; fi
    FUS_LIB_PATH=$lib timeout -k 10 200 python tools/ablate_geom.py 2>&1 | grep "^P=" | sed "s/^/ablate bits ${a:-0}: /"
  done
done | tee $O/ablate_geom.log
