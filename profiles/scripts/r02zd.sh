#!/bin/bash
set -e
O=gpurun_out/r02zd
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/overlap_probe.py --reps 4 --slice 0.05,0.10 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/$O/trace.log; exit 2; }
tail -4 $GRAFT_REPO_ROOT/$O/trace.log
