#!/bin/bash
# mass-apply bench line + launcher tests on the GPU box
set -e
python -m pytest tests/test_bench_launch.py -q -m gpu > gpurun_out/r02z_tests.log 2>&1 || { tail -30 gpurun_out/r02z_tests.log; exit 1; }
tail -2 gpurun_out/r02z_tests.log
python bench.py --mode mass > gpurun_out/r02z_bench_mass.json 2> gpurun_out/r02z_bench_mass.err
cat gpurun_out/r02z_bench_mass.json
FUS_BENCH_FORCE_DIST=1 python bench.py --mode mass --no-cpu-baseline > gpurun_out/r02z_bench_mass_dist_path.json 2>> gpurun_out/r02z_bench_mass.err
cat gpurun_out/r02z_bench_mass_dist_path.json
