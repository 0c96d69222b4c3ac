#!/bin/bash
# round 5: the stream-visibility probe with sub-variants; batch shapes of the in-kernel-geometry kernel (rows / two-row strips / 2x2x5
# with the 20-cell experiment build); solver + operator + bench tests with in-kernel geometry as the solvers' default
O=gpurun_out/r05c
mkdir -p $O
timeout -k 10 400 python tools/stream_visibility_probe.py --reps 6 > $O/probe.log 2>&1; echo "probe rc=$?"; cat $O/probe.log
FUS_LIB_PATH=$PWD/tools/_bin/libfusgpu_cpb20.so timeout -k 10 400 python tools/exp_geom_tiles.py > $O/exp_geom_tiles.log 2>&1; echo "tiles rc=$?"; grep -v amdgpu.ids $O/exp_geom_tiles.log
timeout -k 10 900 python -m pytest tests/test_solver_gpu.py tests/test_operators_gpu.py tests/test_bench_launch.py tests/test_rk4_golden.py tests/test_dolfinx_adaptor.py tests/test_abi.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
echo done
