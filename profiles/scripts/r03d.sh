#!/bin/bash
# round 3: whole GPU suite, fork/join cost probe, default bench line (with aux), 2- and 4-process rehearsal over PEER
set -e
O=gpurun_out/r03d
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 200 python tools/forkjoin_probe.py 50 > $O/forkjoin.log 2>&1 || { tail -20 $O/forkjoin.log; exit 1; }
cat $O/forkjoin.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
cat $O/bench_default.json
FUS_BENCH_REHEARSAL=1 timeout -k 10 400 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_rehearsal2.json 2> $O/bench_rehearsal2.err || { tail -20 $O/bench_rehearsal2.err; exit 1; }
cat $O/bench_rehearsal2.json
