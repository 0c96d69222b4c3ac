#!/bin/bash
set -e
O=gpurun_out/r02ze
mkdir -p $O
run() { echo "== $*"; python tools/overlap_probe.py "$@" 2>&1 | grep "^schedule\|^A \|^operator" ; }
{
run --apply-schedules 4000,6000,8000
run --apply-schedules 4000,6000,8000 --tail
run --permuted --apply-schedules 6000
run --permuted --apply-schedules 6000 --tail
} > $O/schedules.log 2>&1 || { tail -30 $O/schedules.log; exit 1; }
cat $O/schedules.log
