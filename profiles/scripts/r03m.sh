#!/bin/bash
set -e
O=gpurun_out/r03m
mkdir -p $O
timeout -k 10 600 python tools/small_mesh_probe.py > $O/small_mesh.log 2>&1 || { tail -20 $O/small_mesh.log; exit 1; }
cat $O/small_mesh.log
