#!/bin/bash
set -e
O=gpurun_out/r03t
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_halo_gpu.py -x -q -m gpu -k "random_plans" > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
