#!/bin/bash
# round 3: rocprofv3 kernel trace + PMC passes of the shipped library for the headline kernel, the cell mass kernel, the
# linear RK4 step (general G) and the Westervelt step (P = 6)
set -e
bash profiles/run_profile.sh r03_stiffness > gpurun_out/r03j_stiffness.log 2>&1 || { tail -20 gpurun_out/r03j_stiffness.log; exit 1; }
echo stiffness done
bash profiles/run_profile.sh r03_mass --mode mass > gpurun_out/r03j_mass.log 2>&1 || { tail -20 gpurun_out/r03j_mass.log; exit 1; }
echo mass done
bash profiles/run_profile.sh r03_rk4 --mode rk4 --perturbed --steps 10 > gpurun_out/r03j_rk4.log 2>&1 || { tail -20 gpurun_out/r03j_rk4.log; exit 1; }
echo rk4 done
bash profiles/run_profile.sh r03_westervelt --mode westervelt --degree 6 --cells 36 --steps 10 > gpurun_out/r03j_westervelt.log 2>&1 || { tail -20 gpurun_out/r03j_westervelt.log; exit 1; }
echo westervelt done
