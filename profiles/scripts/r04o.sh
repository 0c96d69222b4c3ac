#!/bin/bash
# r04o: per-degree sweep with the round-4 library; the two demos end to end (config-3 size linear box; Westervelt bowl)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
timeout -k 10 500 python tools/sweep.py --degrees 2,3,4,5,6,7,8 > gpurun_out/r04o/sweep_degrees.log 2>&1 || { tail -20 gpurun_out/r04o/sweep_degrees.log; exit 1; }
grep -v Warn gpurun_out/r04o/sweep_degrees.log | tail -30
timeout -k 10 300 python fenicsx-fus-gpu_amd/demo_linear_box.py --cells 54 > gpurun_out/r04o/demo_linear_box_cfg3.log 2>&1 || { tail -20 gpurun_out/r04o/demo_linear_box_cfg3.log; exit 1; }
grep -v Warn gpurun_out/r04o/demo_linear_box_cfg3.log | tail -6
timeout -k 10 300 python fenicsx-fus-gpu_amd/demo_nonlinear_bowl.py --degree 6 --cells 24 --length 0.03 --max-steps 200 > gpurun_out/r04o/demo_nonlinear_bowl.log 2>&1 || { tail -20 gpurun_out/r04o/demo_nonlinear_bowl.log; exit 1; }
grep -v Warn gpurun_out/r04o/demo_nonlinear_bowl.log | tail -6
