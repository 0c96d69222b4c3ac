#!/bin/bash
# round 6, final library: the whole GPU suite, the driver's default command, the driver's N > 1 command rehearsed at FULL size with
# 2 and 4 ranks sharing the one GPU (marked invalid: rehearsal), smoke(), and the degree sweep
O=gpurun_out/r06y
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
for n in 2 4; do
  FUS_BENCH_REHEARSAL=1 timeout -k 10 500 python bench.py --gpus $n --steps 20 --warmup 5 > $O/rehearsal_n${n}_full.json 2> $O/rehearsal_n${n}_full.err; echo "rehearsal n=$n rc=$?"
  grep "harvest\|halo compare\|CHOSEN" $O/rehearsal_n${n}_full.err | cut -c1-260
done
( echo "# tools/sweep.py --degrees 2,3,4,5,6,7,8 (100 back-to-back launches per figure), round-6 library: planned / plan-free stiffness, in-kernel geometry (own contract), mass: gather (default) / static detJ / float-atomic"; timeout -k 10 600 python tools/sweep.py --degrees 2,3,4,5,6,7,8 2>&1 | grep "^P=" ) > $O/sweep_degrees.log; tail -3 $O/sweep_degrees.log | cut -c1-300
