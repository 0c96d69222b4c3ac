#!/bin/bash
# does the exchange run under a chip-filling operator launch?  timings + kernel timeline
set -e
O=gpurun_out/r02za
mkdir -p $O
python tools/overlap_probe.py > $O/overlap_probe.log 2>&1 || { tail -20 $O/overlap_probe.log; exit 1; }
cat $O/overlap_probe.log
python tools/overlap_probe.py --permuted > $O/overlap_probe_permuted.log 2>&1 || { tail -20 $O/overlap_probe_permuted.log; exit 1; }
cat $O/overlap_probe_permuted.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/tools/overlap_probe.py --reps 5 > $GRAFT_REPO_ROOT/$O/trace.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/$O/trace.log; exit 2; }
ls -R $GRAFT_REPO_ROOT/$O/trace | head
