#!/bin/bash
# every-thread preamble up to P = 8: full GPU suite, then ALL rocprofv3 passes again (plan.hpp changed)
set -e
O=gpurun_out/r05z
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -20 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
bash profiles/scripts/r05z_a.sh
bash profiles/scripts/r05z_b.sh
