set -o pipefail
O=gpurun_out/r02m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/west_trace -o west -- python3 $R/bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 > $R/$O/bench_westervelt_P6_traced.json 2> $R/$O/err1 || exit 5
find $R/$O -name "*.db" -delete; find $R/$O -name "*kernel_trace.csv" -delete
cd $R
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r02m/west_trace/west_kernel_stats.csv")))[:6]:
    print("  ", r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
