#!/bin/bash
# default bench line + degree sweep of the round-5 library after the preamble restructure
O=gpurun_out/r05z
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -5 $O/bench_default.err; exit 1; }
tail -c 600 $O/bench_default.json; echo
( echo "# tools/sweep.py --degrees 2,3,4,5,6,7,8 (100 back-to-back launches per figure), round-5 library after the preamble restructure: planned / plan-free stiffness, in-kernel geometry (own contract), mass: gather (default) / static detJ / float-atomic"; timeout -k 10 500 python tools/sweep.py --degrees 2,3,4,5,6,7,8 2>&1 | grep "^P=" ) | tee $O/sweep_degrees.log
