#!/bin/bash
set -e
O=gpurun_out/r03g
mkdir -p $O
timeout -k 10 300 python tools/forkjoin_probe.py 30 > $O/forkjoin.log 2>&1 || { tail -20 $O/forkjoin.log; exit 1; }
cat $O/forkjoin.log
