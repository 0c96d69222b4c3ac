set -o pipefail
O=gpurun_out/r02y
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
for PC in "4 54" "6 36" "2 108"; do set -- $PC
python tools/ab_stiffness.py --degree $1 --cells $2 --dtype f32 --rounds 7 plan plan:0 plan:1 plan:2 runs geom > $O/ab_f32_p$1.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_f32_p$1.log
done
python tools/ab_stiffness.py --degree 4 --rounds 5 plan geom > $O/ab_f64_p4.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_f64_p4.log
python tools/sweep.py --degrees 2,4,6 --dtypes f32 > $O/sweep_f32.log 2>&1 || exit 3
grep -v amdgpu $O/sweep_f32.log
