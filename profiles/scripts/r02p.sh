set -o pipefail
O=gpurun_out/r02p
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python bench.py --mode rk4 --steps 20 --warmup 3 > $O/bench_rk4.json 2> $O/bench_rk4.err || exit 2
python bench.py --mode rk4 --steps 20 --warmup 3 --perturbed > $O/bench_rk4_perturbed.json 2> $O/err2 || exit 2
python bench.py --mode rk4 --steps 20 --warmup 3 --perturbed --in-kernel-geometry > $O/bench_rk4_perturbed_geom.json 2> $O/err3 || exit 2
python bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 > $O/bench_westervelt_P6.json 2> $O/err4 || exit 3
python bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 --in-kernel-geometry > $O/bench_westervelt_P6_geom.json 2> $O/err5 || exit 4
python - <<'PY'
import json
for t in ("bench_rk4","bench_rk4_perturbed","bench_rk4_perturbed_geom","bench_westervelt_P6","bench_westervelt_P6_geom"):
    d=json.loads([l for l in open(f"gpurun_out/r02p/{t}.json") if l.startswith("{")][-1])
    print(t, round(d["ms_per_step"],4), d["config"]["geometry"][:50])
PY
