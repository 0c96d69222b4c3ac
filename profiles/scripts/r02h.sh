set -o pipefail
O=gpurun_out/r02h
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_dolfinx_adaptor.py tests/test_halo_gpu.py -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
for PC in "3 72" "9 24" "10 22"; do set -- $PC
python tools/ab_stiffness.py --degree $1 --cells $2 plan:0 plan:1 plan:2 geom > $O/ab_p$1_f64.log 2>&1 || exit 5
grep -v amdgpu.ids $O/ab_p$1_f64.log
done
python tools/sweep.py --degrees 2,4,6 > $O/sweep.log 2>&1 || { tail $O/sweep.log; exit 6; }
grep -v amdgpu.ids $O/sweep.log
