#!/bin/bash
# round 3: shape of the exchange kernels (threads per workgroup, message elements per workgroup) and stream priority
set -e
O=gpurun_out/r03f
mkdir -p $O
run() { echo "== $*"; timeout -k 10 300 env "$@" python tools/overlap_probe.py --transport peer --paired 5 --reps 30 2>&1 | grep "^paired\|Error\|error" ; }
{
run FUS_IPC_THREADS=256 FUS_IPC_CHUNK=1024
run FUS_IPC_THREADS=256 FUS_IPC_CHUNK=2048
run FUS_IPC_THREADS=256 FUS_IPC_CHUNK=512
run FUS_IPC_THREADS=256 FUS_IPC_CHUNK=256
run FUS_IPC_THREADS=128 FUS_IPC_CHUNK=512
run FUS_IPC_THREADS=64 FUS_IPC_CHUNK=256
run FUS_IPC_THREADS=64 FUS_IPC_CHUNK=512
run FUS_IPC_THREADS=512 FUS_IPC_CHUNK=4096
run FUS_IPC_THREADS=256 FUS_IPC_CHUNK=1024 FUS_COMM_PRIORITY=normal
run FUS_IPC_THREADS=256 FUS_IPC_CHUNK=1024 FUS_IPC_TWO_STREAMS=1
} > $O/sweep.log 2>&1 || { tail -30 $O/sweep.log; exit 1; }
cat $O/sweep.log
