#!/bin/bash
# r04g: vector pass of the RK4 stage with 16-byte accesses + non-temporal streams: solver parity tests, then step times
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04g
timeout -k 10 900 python -m pytest tests/test_solver_gpu.py tests/test_rk4_golden.py tests/test_abi.py -m gpu -x -q > gpurun_out/r04g/tests.log 2>&1
rc=$?
tail -4 gpurun_out/r04g/tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  timeout -k 10 200 python tools/time_rk4.py --no-affine --steps 40 2>&1 | grep "^fused" | sed "s/^/linear general G: /"
  timeout -k 10 200 python tools/time_rk4.py --steps 40 2>&1 | grep "^fused" | sed "s/^/linear affine: /"
  timeout -k 10 200 python tools/time_rk4.py --westervelt --degree 6 --cells 36 --steps 20 2>&1 | grep "^fused" | sed "s/^/westervelt P6: /"
done | tee gpurun_out/r04g/step_times.log
