#!/bin/bash
# round 3: the judged comparison (apply with both exchanges vs ONE launch), paired rounds, per transport; halo tests first
set -e
O=gpurun_out/r03e
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_halo_gpu.py -x -q -m gpu > $O/pytest_halo.log 2>&1 || { tail -40 $O/pytest_halo.log; exit 1; }
tail -2 $O/pytest_halo.log
run() { timeout -k 10 300 python tools/overlap_probe.py "$@" 2>&1 | grep "^paired\|Error\|error" ; }
{
run --transport peer --paired 7 --reps 40
run --transport peer --permuted --paired 7 --reps 40
run --transport local --paired 7 --reps 40
run --transport native --paired 7 --reps 40
} > $O/paired.log 2>&1 || { tail -30 $O/paired.log; exit 1; }
cat $O/paired.log
