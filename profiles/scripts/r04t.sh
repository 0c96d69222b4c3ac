#!/bin/bash
# atomic-free mass apply (csrc/mass_gather.hpp): GPU suite, default bench line, rocprofv3 passes of --mode mass (gather kernel)
# and --mode mass --mass-atomic (batch plan + float atomics) with the same library
set -e
O=gpurun_out/r04t
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
bash profiles/run_profile.sh r04t_mass_gather --mode mass > $O/prof_mass.log 2>&1 || { tail -20 $O/prof_mass.log; exit 1; }
bash profiles/run_profile.sh r04t_mass_atomic --mode mass --mass-atomic > $O/prof_mass_atomic.log 2>&1 || { tail -20 $O/prof_mass_atomic.log; exit 1; }
echo profiles done
