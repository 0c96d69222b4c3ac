#!/bin/bash
set -e
O=gpurun_out/r03k
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_solver_gpu.py tests/test_bench_launch.py tests/test_abi.py tests/test_dolfinx_adaptor.py tests/test_resource_usage.py -x -q -m gpu > $O/pytest_gpu2.log 2>&1 || { tail -60 $O/pytest_gpu2.log; exit 1; }
tail -3 $O/pytest_gpu2.log
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
python -c "
import json; d=json.load(open('$O/bench_default.json'))
print('headline', d['value']/1e9, d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'])
for k,v in d['aux'].items(): print(k, v and (v['value']/1e9, v['ms_per_step'], v['roofline']['frac']))
print('cpu', d['cpu_baseline']['value']/1e9, d['cpu_baseline']['cores'], d['cpu_baseline']['single_thread_value']/1e9, d['cpu_baseline']['noisy'])"
timeout -k 10 300 python bench.py --mode mass_diag --no-cpu-baseline > $O/bench_mass_diag.json 2> $O/bench_mass_diag.err || { tail -20 $O/bench_mass_diag.err; exit 1; }
cat $O/bench_mass_diag.json | head -c 600; echo
