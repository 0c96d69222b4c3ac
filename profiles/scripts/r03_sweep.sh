#!/bin/bash
set -e
O=gpurun_out/r03_sweep
mkdir -p $O
timeout -k 10 600 python tools/sweep.py --degrees 2,3,4,5,6,7,8 --dtypes f64,f32 --reps 20 > $O/sweep.log 2>&1 || { tail -20 $O/sweep.log; exit 1; }
cat $O/sweep.log | grep -v amdgpu.ids
