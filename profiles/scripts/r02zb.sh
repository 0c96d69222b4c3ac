#!/bin/bash
# mitigations for the starved exchange kernel: slicing the operator launch, reserved CUs, fewer RCCL channels
set -e
O=gpurun_out/r02zb
mkdir -p $O
run() { echo "== $*"; python tools/overlap_probe.py "$@" 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|version\|Hostname\|Librccl" ; }
{
run --slice 0.1
run --slice 0.25
run --max-channels 4 --slice 0.1
run --reserve-cus 8 --max-channels 8
run --reserve-cus 8 --mask-style spread --max-channels 8
run --reserve-cus 16 --mask-style spread --max-channels 8
run --reserve-cus 32 --mask-style spread
} > $O/mitigations.log 2>&1 || { tail -30 $O/mitigations.log; exit 1; }
cat $O/mitigations.log
