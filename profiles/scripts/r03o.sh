#!/bin/bash
set -e
O=gpurun_out/r03o
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_abi.py -x -q -m gpu 2>&1 | tail -2
run() { echo "== $*"; timeout -k 10 300 python tools/overlap_probe.py "$@" 2>&1 | grep "^paired\|Error\|error" ; }
{
run --transport peer --paired 7 --reps 40
run --transport peer --paired 7 --reps 40 --message-scale 0.01
run --transport peer --paired 7 --reps 40 --message-scale 0.25
} > $O/paired_scale.log 2>&1 || { tail -30 $O/paired_scale.log; exit 1; }
cat $O/paired_scale.log
