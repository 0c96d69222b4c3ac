#!/bin/bash
# kernel timeline of the concurrent schedule over the PEER transport (self-neighbour world, config-4-size messages)
set -e
O=$(pwd)/gpurun_out/r03c
mkdir -p $O
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 $R/tools/overlap_probe.py --transport peer --apply-schedules 0 --reps 12 > $O/probe.log 2>&1 || { tail -20 $O/probe.log; exit 1; }
cd $R
grep "^schedule\|^A " $O/probe.log
python tools/timeline.py $O/trace 400 > $O/timeline.txt
find $O -name "*.db" -delete
wc -l $O/timeline.txt
