#!/bin/bash
# the round-end sequence: whole GPU suite, smoke, default bench line
set -e
O=gpurun_out/r03r
mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1 || { tail -60 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench_default.json'))
print('headline', d['value']/1e9, d['roofline']['frac'], d['config']['lib_built_from_tree'], d['roofline']['traffic_source'][:60])
for k,v in d['aux'].items(): print(k, v and (v['value']/1e9, v['ms_per_step'], v['roofline']['frac']))"
