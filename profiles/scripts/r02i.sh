set -o pipefail
O=gpurun_out/r02i
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 > $O/bench_westervelt_P6.json 2> $O/bench_westervelt_P6.err || exit 3
python bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 --in-kernel-geometry > $O/bench_westervelt_P6_geom.json 2> $O/bench_westervelt_P6_geom.err || exit 4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/west_trace -o west -- python3 $R/bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 > $R/$O/bench_westervelt_P6_traced.json 2> $R/$O/bench_westervelt_P6_traced.err || exit 5
find $R/$O -name "*.db" -delete; find $R/$O -name "*kernel_trace.csv" -delete
cd $R
python - <<'PY'
import json, csv
for t in ("bench_westervelt_P6","bench_westervelt_P6_geom"):
    d=json.loads([l for l in open(f"gpurun_out/r02i/{t}.json") if l.startswith("{")][-1])
    print(t, d["ms_per_step"], d["config"]["geometry"])
for r in list(csv.DictReader(open("gpurun_out/r02i/west_trace/west_kernel_stats.csv")))[:4]:
    print("  ", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
