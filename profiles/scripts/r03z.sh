#!/bin/bash
set -e
O=gpurun_out/r03z
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bench_launch.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
