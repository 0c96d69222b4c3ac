#!/bin/bash
# r04b: the default bench line with the new aux entries, --mode scatter, the config-gap tests
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04b
timeout -k 10 600 python bench.py --steps 20 --warmup 10 > gpurun_out/r04b/bench_default.json 2> gpurun_out/r04b/bench_default.err || { tail -30 gpurun_out/r04b/bench_default.err; exit 1; }
timeout -k 10 300 python bench.py --mode scatter --steps 100 > gpurun_out/r04b/bench_scatter.json 2> gpurun_out/r04b/bench_scatter.err || { tail -30 gpurun_out/r04b/bench_scatter.err; exit 1; }
timeout -k 10 900 python -m pytest tests/test_operators_gpu.py tests/test_solver_gpu.py -m gpu -x -q -k "config1 or config3 or config5" > gpurun_out/r04b/gap_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04b/gap_tests.log
exit $rc
