#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the FINAL round-3 library: headline kernel (default bench command, --no-aux) and cell mass
set -e
bash profiles/run_profile.sh r03_final > gpurun_out/r03_final.log 2>&1 || { tail -20 gpurun_out/r03_final.log; exit 1; }
echo stiffness done
bash profiles/run_profile.sh r03_final_mass --mode mass > gpurun_out/r03_final_mass.log 2>&1 || { tail -20 gpurun_out/r03_final_mass.log; exit 1; }
echo mass done
