#!/bin/bash
# headline passes once more (the kernel-trace pass of r05z_g.sh caught a 1.6 ms outlier launch inside its timed region)
set -e
O=gpurun_out/r05z
mkdir -p $O
bash profiles/run_profile.sh r05z > $O/prof_r05z.log 2>&1 || { tail -20 $O/prof_r05z.log; exit 1; }
echo "r05z done"
