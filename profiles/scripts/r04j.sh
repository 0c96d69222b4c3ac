#!/bin/bash
# r04j: after the streaming stores: whole GPU suite, the default bench line, rocprofv3 passes of the RK4 step (traffic per kernel)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04j
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04j/pytest_gpu.log 2>&1
rc=$?
tail -3 gpurun_out/r04j/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r04j/bench_default.json 2> gpurun_out/r04j/bench_default.err || { tail -20 gpurun_out/r04j/bench_default.err; exit 1; }
bash profiles/run_profile.sh r04_final_rk4 --mode rk4 --perturbed > gpurun_out/r04j/prof_rk4.log 2>&1 || { tail -20 gpurun_out/r04j/prof_rk4.log; exit 1; }
bash profiles/run_profile.sh r04_final_rk4_geom --mode rk4 --perturbed --in-kernel-geometry > gpurun_out/r04j/prof_rk4_geom.log 2>&1 || { tail -20 gpurun_out/r04j/prof_rk4_geom.log; exit 1; }
echo done
