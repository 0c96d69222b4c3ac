#!/bin/bash
set -e
O=gpurun_out/r02zg
mkdir -p $O
run() { echo "== $*"; python tools/overlap_probe.py "$@" 2>&1 | grep "^schedule\|^A \|^operator" ; }
{
run --apply-schedules 6000,8000,9000,9500
run --apply-schedules 7000,8500,9200,10000
} > $O/lead_sizes.log 2>&1 || { tail -30 $O/lead_sizes.log; exit 1; }
cat $O/lead_sizes.log
