#!/bin/bash
# round 3, first call: primitives of the IPC halo transport between two processes on one GPU; exposed cost of an exchange
# made of small kernels (the library's in-process transport) next to the operator; boundary cells on a second stream
set -e
O=gpurun_out/r03a
mkdir -p $O
timeout -k 10 280 tools/_bin/ipc_probe 200 > $O/ipc_probe.log 2>&1 || { echo "ipc_probe rc=$?"; tail -30 $O/ipc_probe.log; }
cat $O/ipc_probe.log
run() { echo "== $*"; timeout -k 10 300 python tools/overlap_probe.py "$@" 2>&1 | grep "^schedule\|^A \|^operator\|^two-stream\|^local\|^native\|Error\|error" ; }
{
run --transport local --apply-schedules 6000 --two-stream
run --transport local --permuted --apply-schedules 6000
} > $O/overlap_local.log 2>&1 || { tail -30 $O/overlap_local.log; exit 1; }
cat $O/overlap_local.log
