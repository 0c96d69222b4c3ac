set -o pipefail
O=gpurun_out/r02_final3
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail $O/smoke.log; exit 9; }
tail -1 $O/smoke.log
bash profiles/run_profile.sh r02_final3 > $O/run_profile.log 2>&1 || { tail -20 $O/run_profile.log; exit 2; }
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 3
python bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err || exit 3
python bench.py --mode stiffness_geom --no-cpu-baseline > $O/bench_geom.json 2> $O/bench_geom.err || exit 4
FUS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline > $O/bench_forcedist_native.json 2> $O/bench_forcedist.err || exit 5
python -c "
import json
for t in ('bench_default','bench_f32','bench_geom','bench_forcedist_native'):
    d=json.loads([l for l in open('$O/'+t+'.json') if l.startswith('{')][-1]); r=d['roofline']; print(t, d['value'], round(d['ms_per_step'],4), r['frac'], r['isolated_frac'], r['traffic_source'][:40], d['config'].get('halo_exposed_ms'))"
