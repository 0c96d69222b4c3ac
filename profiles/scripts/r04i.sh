#!/bin/bash
# r04i: streaming (non-temporal) stores in the vector kernels: parity tests, the interleave probe, RK4 step times with the
# tuning knob off (FUS_TUNE_VECTOR_STREAM = 0: plain stores, the round-3 behaviour) and on (auto)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04i
timeout -k 10 900 python -m pytest tests/test_solver_gpu.py tests/test_rk4_golden.py tests/test_operators_gpu.py -m gpu -x -q -k "not maximum_size" > gpurun_out/r04i/tests.log 2>&1
rc=$?
tail -4 gpurun_out/r04i/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python tools/interleave_probe.py 2>&1 | grep "^round" | tee gpurun_out/r04i/interleave_probe_streaming_stores.log
for i in 1 2; do
  for m in 0 1; do
    echo "FUS_TUNE_VECTOR_STREAM=$m (round $i)"
    FUS_VECTOR_STREAM=$m timeout -k 10 200 python tools/time_rk4.py --no-affine --steps 40 2>&1 | grep "^fused" | sed "s/^/  linear general G: /"
    FUS_VECTOR_STREAM=$m timeout -k 10 200 python tools/time_rk4.py --steps 40 2>&1 | grep "^fused" | sed "s/^/  linear affine: /"
    FUS_VECTOR_STREAM=$m timeout -k 10 200 python tools/time_rk4.py --westervelt --degree 6 --cells 36 --steps 20 2>&1 | grep "^fused" | sed "s/^/  westervelt P6: /"
  done
done | tee gpurun_out/r04i/step_times.log
