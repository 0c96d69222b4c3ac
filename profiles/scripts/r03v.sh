#!/bin/bash
set -e
O=gpurun_out/r03v
mkdir -p $O
run() { echo "== $*"; timeout -k 10 300 python tools/overlap_probe.py "$@" 2>&1 | grep "^paired\|Error\|error" ; }
{
run --transport ipc --paired 9 --reps 40
run --transport ipc --paired 9 --reps 40 --permuted
run --transport native --paired 5 --reps 40
} > $O/paired_final.log 2>&1 || { tail -30 $O/paired_final.log; exit 1; }
cat $O/paired_final.log
