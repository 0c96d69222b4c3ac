#!/bin/bash
set -e
O=gpurun_out/r03h
mkdir -p $O
run() { echo "== $*"; timeout -k 10 300 python tools/overlap_probe.py "$@" 2>&1 | grep "^paired\|Error\|error" ; }
{
run --transport peer --paired 7 --reps 40
run --transport peer --paired 7 --reps 40 --random-indices
run --transport peer --paired 7 --reps 40 --permuted
run --transport native --paired 7 --reps 40
} > $O/paired.log 2>&1 || { tail -30 $O/paired.log; exit 1; }
cat $O/paired.log
