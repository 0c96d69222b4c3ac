#!/bin/bash
# round 6, final library: the two demos end to end (config-3 size linear box; Westervelt bowl), and the config-3 RK4 loop from the plain
# C++ host over the C ABI (affine fast path; in-kernel geometry and general G on warped cells)
O=gpurun_out/r06x
mkdir -p $O
timeout -k 10 300 python fenicsx-fus-gpu_amd/demo_linear_box.py --cells 54 > $O/demo_linear_box_cfg3.log 2>&1 || { tail -20 $O/demo_linear_box_cfg3.log; exit 1; }
grep -v "Warn\|amdgpu.ids" $O/demo_linear_box_cfg3.log | tail -4
timeout -k 10 300 python fenicsx-fus-gpu_amd/demo_nonlinear_bowl.py --degree 6 --cells 24 --length 0.03 --max-steps 200 > $O/demo_nonlinear_bowl.log 2>&1 || { tail -20 $O/demo_nonlinear_bowl.log; exit 1; }
grep -v "Warn\|amdgpu.ids" $O/demo_nonlinear_bowl.log | tail -4
make -C examples > /dev/null 2>&1
for g in "0 0" "2 1" "1 1"; do timeout -k 10 300 ./examples/c_abi_linear_box 4 54 840 $g 2>&1 | tail -3; done | tee $O/c_abi_linear_box.log
