#!/bin/bash
# r04h: the whole GPU suite on the round-4 library, then the full-size rehearsals (profiles/scripts/r04e.sh)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r04h/pytest_gpu.log 2>&1
rc=$?
tail -4 gpurun_out/r04h/pytest_gpu.log
exit $rc
