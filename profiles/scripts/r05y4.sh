#!/bin/bash
for v in tree rcp tree rcp; do
  if [ $v = tree ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_rcp.so; fi
  FUS_LIB_PATH=$lib timeout -k 10 200 python tools/ablate_geom.py 2>&1 | grep -E "^P=|one apply" | sed "s/^/$v: /"
done
