set -o pipefail
O=gpurun_out/r02o
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
for PC in "4 54" "6 36" "2 108" "3 72" "5 43" "7 30"; do set -- $PC
python tools/ab_stiffness.py --degree $1 --cells $2 --rounds 7 plan raw geom > $O/ab_runs_p$1.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_runs_p$1.log
done
python tools/ab_stiffness.py --degree 4 --dtype f32 --rounds 7 plan raw runs geom > $O/ab_runs_p4_f32.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_runs_p4_f32.log
python tools/ab_stiffness.py --degree 6 --cells 36 --dtype f32 --rounds 7 plan raw runs > $O/ab_runs_p6_f32.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_runs_p6_f32.log
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err || exit 3
python -c "
import json; d=json.loads([l for l in open('$O/bench_default.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['isolated_frac'])"
