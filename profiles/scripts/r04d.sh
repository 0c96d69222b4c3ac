#!/bin/bash
# r04d: exclusive-dof marks in the batch plan: parity test, then A/B of the cell mass apply per degree
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04d
timeout -k 10 600 python -m pytest tests/test_operators_gpu.py -m gpu -x -q -k "exclusive or mass" > gpurun_out/r04d/tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04d/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python tools/ab_mass_exclusive.py > gpurun_out/r04d/ab_mass_exclusive.log 2>&1 || { tail -20 gpurun_out/r04d/ab_mass_exclusive.log; exit 1; }
grep "^P=" gpurun_out/r04d/ab_mass_exclusive.log
