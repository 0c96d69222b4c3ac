set -o pipefail
O=gpurun_out/r02k
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
bash profiles/run_profile.sh r02k > $O/run_profile.log 2>&1 || { tail -20 $O/run_profile.log; exit 2; }
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 3
python bench.py --mode stiffness_geom --no-cpu-baseline > $O/bench_geom.json 2> $O/bench_geom.err || exit 4
FUS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --halo native > $O/bench_forcedist_native.json 2> $O/bench_forcedist_native.err || exit 5
python bench.py --mode rk4 --steps 20 --warmup 3 > $O/bench_rk4.json 2> $O/bench_rk4.err || exit 6
python tools/host_overhead.py > $O/host_overhead.log 2>&1 || exit 7
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/west_trace -o west -- python3 $R/bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 --in-kernel-geometry > $R/$O/bench_westervelt_P6_geom_traced.json 2> $R/$O/bench_westervelt_P6_geom_traced.err || exit 8
find $R/$O -name "*.db" -delete; find $R/$O -name "*kernel_trace.csv" -delete
cd $R
python - <<'PY'
import json, csv
for t in ("bench_default","bench_geom","bench_forcedist_native","bench_rk4"):
    d=json.loads([l for l in open(f"gpurun_out/r02k/{t}.json") if l.startswith("{")][-1])
    r=d.get("roofline") or {}
    print(t, "ms/step", round(d["ms_per_step"],4), "value", d["value"], "frac", r.get("frac"), "isolated", r.get("isolated_frac"), "exposed", d["config"].get("halo_exposed_ms"))
for r in list(csv.DictReader(open("gpurun_out/r02k/west_trace/west_kernel_stats.csv")))[:3]:
    print("  ", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
grep -v amdgpu $O/host_overhead.log | grep -v "^\[W" | tail -9
