#!/bin/bash
set -e
O=gpurun_out/r03n
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_halo_gpu.py tests/test_rk4_golden.py tests/test_solver_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python fenicsx-fus-gpu_amd/time_operators.py --degree 4 --cells 25 --nreps 20 2>&1 | tail -5
