#!/bin/bash
set -e
O=gpurun_out/r03x
mkdir -p $O
BASE=$(pwd)/fenicsx-fus-gpu_amd/csrc/_ab/libfusgpu_base.so
{
for i in 1 2; do
  echo "== base";  FUS_LIB_PATH=$BASE timeout -k 10 200 python tools/small_mesh_probe.py 16,22,25,29 2>&1 | grep "^N="
  echo "== new";   timeout -k 10 200 python tools/small_mesh_probe.py 16,22,25,29 2>&1 | grep "^N="
done
} > $O/ab_small.log 2>&1 || { tail -30 $O/ab_small.log; exit 1; }
cut -c1-175 $O/ab_small.log
