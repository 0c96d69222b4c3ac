set -o pipefail
O=gpurun_out/r02f
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python tools/ab_stiffness.py --degree 4 plan geom > $O/ab_p4_geom.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_p4_geom.log
python tools/ab_stiffness.py --degree 5 --cells 43 plan geom > $O/ab_p5_geom.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_p5_geom.log
python tools/ab_stiffness.py --degree 2 --cells 108 plan geom > $O/ab_p2_geom.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_p2_geom.log
python bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 > $O/bench_westervelt_P6.json 2> $O/bench_westervelt_P6.err || exit 3
python bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 --in-kernel-geometry > $O/bench_westervelt_P6_geom.json 2> $O/bench_westervelt_P6_geom.err || exit 4
python - <<'PY'
import json
for t in ("","_geom"):
    d=json.loads([l for l in open(f"gpurun_out/r02f/bench_westervelt_P6{t}.json") if l.startswith("{")][-1])
    print("westervelt", t, d["ms_per_step"], d["config"]["geometry"])
PY
