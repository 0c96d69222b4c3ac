set -o pipefail
O=gpurun_out/r02t
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 9; }
tail -1 $O/smoke.log
bash profiles/run_profile.sh r02t > $O/run_profile.log 2>&1 || { tail -20 $O/run_profile.log; exit 2; }
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 3
python -c "
import json
d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['isolated_frac'], r['traffic'], r['traffic_source']); print(d['cpu_baseline'])"
