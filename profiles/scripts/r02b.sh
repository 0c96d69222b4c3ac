set -o pipefail
O=gpurun_out/r02b
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
bash profiles/run_profile.sh r02b > $O/run_profile.log 2>&1 || { tail -20 $O/run_profile.log; exit 2; }
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 3
cut -c1-400 $O/bench_default.json
python tools/ab_stiffness.py --degree 4 --isolated plan:0 plan:1 plan:2 geom > $O/ab_p4_f64_isolated.log 2>&1 || exit 4
cat $O/ab_p4_f64_isolated.log
for P in 5 7 8; do
python tools/ab_stiffness.py --degree $P --cells $((216/P)) plan:0 plan:1 plan:2 geom > $O/ab_p${P}_f64.log 2>&1 || exit 5
cat $O/ab_p${P}_f64.log
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/west_trace -o west -- python3 $R/bench.py --mode westervelt --degree 6 --cells 36 --steps 20 --warmup 3 > $R/$O/bench_westervelt_P6.json 2> $R/$O/bench_westervelt_P6.err || exit 6
cat $R/$O/bench_westervelt_P6.json | cut -c1-300
head -8 $R/$O/west_trace/*/west_kernel_stats.csv 2>/dev/null || find $R/$O/west_trace -name "*stats*"
find $R/$O/west_trace -name "*.db" -delete
find $R/$O/west_trace -name "*kernel_trace.csv" -delete
