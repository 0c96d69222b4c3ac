#!/bin/bash
set -e
O=gpurun_out/r03u
mkdir -p $O
timeout -k 10 700 python -m pytest tests/test_halo_gpu.py -x -q -m gpu -k "soak" --durations=2 > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -6 $O/pytest.log
