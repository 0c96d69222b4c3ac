#!/bin/bash
set -e
O=gpurun_out/r02zf
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/overlap_probe.py --reps 3 --apply-schedules 6000 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/$O/trace.log; exit 2; }
grep "^schedule" $GRAFT_REPO_ROOT/$O/trace.log
