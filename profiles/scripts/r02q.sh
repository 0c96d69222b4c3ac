set -o pipefail
O=gpurun_out/r02q
mkdir -p $O
bash profiles/run_profile.sh r02q > $O/run_profile.log 2>&1 || { tail -20 $O/run_profile.log; exit 2; }
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 3
python bench.py --mode stiffness_geom --no-cpu-baseline > $O/bench_geom.json 2> $O/bench_geom.err || exit 4
python -c "
import json
for t in ('bench_default','bench_geom'):
    d=json.loads([l for l in open('$O/'+t+'.json') if l.startswith('{')][-1]); r=d['roofline']; print(t, d['value'], d['ms_per_step'], r['frac'], r['isolated_frac'], r['traffic'], r['traffic_source'])"
