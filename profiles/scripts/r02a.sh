set -o pipefail
O=gpurun_out/r02a
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
tail -1 $O/bench_default.json | cut -c1-600
python bench.py --mode stiffness_geom --no-cpu-baseline > $O/bench_geom.json 2> $O/bench_geom.err || exit 2
python tools/ab_stiffness.py --degree 4 plan:0 plan:1 plan:2 geom > $O/ab_p4_f64.log 2>&1 || exit 3
cat $O/ab_p4_f64.log
python tools/ab_stiffness.py --degree 6 --cells 36 plan:0 plan:1 plan:2 geom > $O/ab_p6_f64.log 2>&1 || exit 4
cat $O/ab_p6_f64.log
python tools/ab_stiffness.py --degree 4 --dtype f32 plan:30 plan:0 plan:1 plan:2 geom > $O/ab_p4_f32.log 2>&1 || exit 5
cat $O/ab_p4_f32.log
FUS_LIB_PATH=$PWD/fenicsx-fus-gpu_amd/csrc/_ab/libfusgpu_slp.so python tools/ab_stiffness.py --degree 4 --dtype f32 plan:30 plan:0 plan:1 plan:2 geom > $O/ab_p4_f32_slp.log 2>&1 || exit 6
cat $O/ab_p4_f32_slp.log
python tools/ab_stiffness.py --degree 6 --cells 36 --dtype f32 plan:0 plan:1 plan:2 geom > $O/ab_p6_f32.log 2>&1 || exit 7
cat $O/ab_p6_f32.log
FUS_LIB_PATH=$PWD/fenicsx-fus-gpu_amd/csrc/_ab/libfusgpu_slp.so python tools/ab_stiffness.py --degree 6 --cells 36 --dtype f32 plan:0 plan:1 plan:2 > $O/ab_p6_f32_slp.log 2>&1 || exit 8
cat $O/ab_p6_f32_slp.log
