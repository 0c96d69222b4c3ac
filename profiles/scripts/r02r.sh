set -o pipefail
O=gpurun_out/r02r
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_operators_gpu.py -m gpu -x -q -k "planned_kernel_builds or all_degrees" > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
for PC in "8 27" "9 24" "10 22"; do set -- $PC
python tools/ab_stiffness.py --degree $1 --cells $2 --rounds 7 plan:0 plan:1 plan:2 geom > $O/ab_p$1.log 2>&1 || exit 2
grep -v amdgpu.ids $O/ab_p$1.log
done
