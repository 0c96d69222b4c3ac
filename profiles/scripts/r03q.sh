#!/bin/bash
set -e
O=gpurun_out/r03q
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_rk4_golden.py tests/test_abi.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
