set -o pipefail
O=gpurun_out/r02g
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python bench.py --mode rk4 --steps 20 --warmup 3 > $O/bench_rk4.json 2> $O/bench_rk4.err || exit 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/west_trace -o west -- python3 $R/bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 --in-kernel-geometry > $R/$O/bench_westervelt_P6_geom.json 2> $R/$O/bench_westervelt_P6_geom.err || exit 3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/rk4_trace -o rk4 -- python3 $R/bench.py --mode rk4 --steps 20 --warmup 3 > $R/$O/bench_rk4_traced.json 2> $R/$O/bench_rk4_traced.err || exit 4
find $R/$O -name "*.db" -delete; find $R/$O -name "*kernel_trace.csv" -delete
cd $R
python - <<'PY'
import json, csv
for t in ("bench_rk4","bench_westervelt_P6_geom"):
    d=json.loads([l for l in open(f"gpurun_out/r02g/{t}.json") if l.startswith("{")][-1])
    print(t, d["ms_per_step"], d["config"]["geometry"])
for t in ("west_trace/west","rk4_trace/rk4"):
    for r in list(csv.DictReader(open(f"gpurun_out/r02g/{t}_kernel_stats.csv")))[:5]:
        print("  ", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
