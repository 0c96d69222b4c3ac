set -o pipefail
O=gpurun_out/r02v
mkdir -p $O
python fenicsx-fus-gpu_amd/demo_linear_box.py --degree 4 --cells 54 > $O/demo_linear_box_cfg3.log 2>&1 || { tail $O/demo_linear_box_cfg3.log; exit 1; }
grep -v amdgpu $O/demo_linear_box_cfg3.log | tail -6
python fenicsx-fus-gpu_amd/demo_linear_box.py --degree 4 --cells 54 --reference-sequence > $O/demo_linear_box_cfg3_refseq.log 2>&1 || exit 1
grep -v amdgpu $O/demo_linear_box_cfg3_refseq.log | tail -3
python fenicsx-fus-gpu_amd/time_operators.py --degree 2 --cells 18 > $O/time_operators_cfg1.log 2>&1 || exit 2
grep -v amdgpu $O/time_operators_cfg1.log | tail -5
python fenicsx-fus-gpu_amd/time_operators.py --degree 4 --cells 25 > $O/time_operators_cfg2.log 2>&1 || exit 2
grep -v amdgpu $O/time_operators_cfg2.log | tail -5
