#!/bin/bash
set -e
O=gpurun_out/r03i
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_halo_gpu.py tests/test_solver_gpu.py tests/test_dolfinx_adaptor.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run() { echo "== $*"; timeout -k 10 300 env $ENVV python tools/overlap_probe.py "$@" 2>&1 | grep "^paired\|Error\|error" ; }
{
run --transport peer --paired 7 --reps 40
ENVV="FUS_HALO_EVENT_SYNC=1" run --transport peer --paired 7 --reps 40
run --transport peer --paired 7 --reps 40 --permuted
run --transport local --paired 7 --reps 40
} > $O/paired.log 2>&1 || { tail -30 $O/paired.log; exit 1; }
cat $O/paired.log
