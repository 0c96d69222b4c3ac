#!/bin/bash
# round 3: PEER halo transport -- parity tests (in-process ranks and real processes sharing the GPU), then what an
# exchange costs next to the operator for each transport and schedule
set -e
O=gpurun_out/r03b
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_halo_gpu.py -x -q -m gpu > $O/pytest_halo.log 2>&1 || { tail -40 $O/pytest_halo.log; exit 1; }
tail -3 $O/pytest_halo.log
run() { echo "== $*"; timeout -k 10 300 python tools/overlap_probe.py "$@" 2>&1 | grep "^schedule\|^A \|^operator\|^two-stream\|^local\|^native\|^peer\|Error\|error" ; }
{
run --transport peer --apply-schedules 6000
run --transport peer --permuted --apply-schedules 6000
run --transport local --apply-schedules 6000
run --transport native --apply-schedules 6000
} > $O/overlap.log 2>&1 || { tail -30 $O/overlap.log; exit 1; }
cat $O/overlap.log
