#!/bin/bash
set -e
O=gpurun_out/r02zc
mkdir -p $O
run() { echo "== $*"; python tools/overlap_probe.py "$@" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|version\|Hostname\|Librccl" ; }
{
run --slice 0.05
run --slice 0.05,0.10
run --slice 0.03,0.06
run --slice 0.05,0.05,0.05
run --permuted --slice 0.05,0.10
run --permuted --slice 0.05,0.05,0.05
} > $O/slices.log 2>&1 || { tail -30 $O/slices.log; exit 1; }
cat $O/slices.log
