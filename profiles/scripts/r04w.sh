#!/bin/bash
# final passes of the round's library: mass (atomic-free kernel) counters for traffic_latest.json, default bench line
set -e
O=gpurun_out/r04w
mkdir -p $O
bash profiles/run_profile.sh r04w_mass_gather --mode mass > $O/prof_mass.log 2>&1 || { tail -20 $O/prof_mass.log; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
echo done
