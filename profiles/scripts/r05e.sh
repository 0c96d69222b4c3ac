#!/bin/bash
# round 5: mass apply with detJ in row order (static companion) + row-subset plans: tests; batch shapes of the in-kernel-geometry kernel
# under sustained load; TCC_EA0_ATOMIC of that kernel with the row-ordered and the strip-ordered plan; the default bench line
O=gpurun_out/r05e
mkdir -p $O
REPO=$PWD
timeout -k 10 600 python -m pytest tests/test_operators_gpu.py tests/test_bench_launch.py tests/test_solver_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
FUS_LIB_PATH=$PWD/tools/_bin/libfusgpu_cpb20.so timeout -k 10 400 python tools/exp_geom_tiles.py > $O/exp_geom_tiles.log 2>&1; echo "tiles rc=$?"; grep -v amdgpu.ids $O/exp_geom_tiles.log
cd /tmp && export TMPDIR=/tmp
for s in 0 1; do
  FUS_PLAN_STRIP_ORDER=$s timeout -k 10 300 rocprofv3 --pmc TCC_EA0_ATOMIC_sum --kernel-trace --output-format csv -d $REPO/$O/strips$s -o tcc -- python3 $REPO/bench.py --mode stiffness_geom --steps 20 --warmup 3 --no-cpu-baseline --no-check > $REPO/$O/bench_strips$s.json 2> $REPO/$O/bench_strips$s.err || echo "pmc pass strips=$s failed"
done
cd $REPO
python - <<'PY'
import csv, glob, json
for s in (0, 1):
    fs = glob.glob(f"gpurun_out/r05e/strips{s}/**/*counter_collection.csv", recursive=True)
    vals = [float(r["Counter_Value"]) for f in fs for r in csv.DictReader(open(f)) if "stiffness_plan_geom_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "TCC_EA0_ATOMIC_sum"]
    ts = glob.glob(f"gpurun_out/r05e/strips{s}/**/*kernel_trace.csv", recursive=True)
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for f in ts for r in csv.DictReader(open(f)) if "stiffness_plan_geom_kernel" in r["Kernel_Name"]]
    print(json.dumps({"strip_order": s, "launches": len(vals), "TCC_EA0_ATOMIC_per_launch": sum(vals) / max(len(vals), 1), "mean_dispatch_us_under_profiler": sum(dur) / max(len(dur), 1) / 1e3}))
PY
find $O -name "*.db" -delete
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05e/bench_default.json"))
print(json.dumps(d["roofline"].get("secondary"), indent=None))
print(json.dumps({k: d["aux"]["mass"]["roofline"].get(k) for k in ("kernel_ms", "frac", "static_detJ_kernel_ms", "static_detJ_frac", "atomic_kernel_ms")}))
print(json.dumps({k: d["aux"]["stiffness_in_kernel_geometry"]["roofline"].get(k) for k in ("kernel_ms", "kernel_ms_first_burst", "frac")}))
PY
echo done
