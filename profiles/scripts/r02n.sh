set -o pipefail
O=gpurun_out/r02n
mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
python tools/ab_stiffness.py --degree 4 --rounds 9 plan raw:-1 geom > $O/ab_runs_p4.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_runs_p4.log
python tools/ab_stiffness.py --degree 6 --cells 36 --rounds 9 plan raw:-1 geom > $O/ab_runs_p6.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_runs_p6.log
python tools/ab_stiffness.py --degree 2 --cells 108 --rounds 9 plan raw:-1 > $O/ab_runs_p2.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_runs_p2.log
python tools/ab_stiffness.py --degree 4 --dtype f32 --rounds 9 plan raw:-1 > $O/ab_runs_p4_f32.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_runs_p4_f32.log
python tools/ab_stiffness.py --degree 8 --cells 27 --rounds 9 plan raw:-1 > $O/ab_runs_p8.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_runs_p8.log
