#!/bin/bash
# round 5: partitioned mass apply on the atomic-free kernel (row split), the repaired fork / join contract test, solver set-up on the default
# mass operator; where the 0.155 vs 0.184 ms of the in-kernel-geometry kernel comes from; batch shapes under sustained load
O=gpurun_out/r05d
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_halo_gpu.py tests/test_operators_gpu.py tests/test_solver_gpu.py tests/test_distributed.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout -k 10 300 python tools/geom_variance_probe.py > $O/geom_variance_probe.log 2>&1; echo "variance rc=$?"; grep -v amdgpu.ids $O/geom_variance_probe.log
FUS_LIB_PATH=$PWD/tools/_bin/libfusgpu_cpb20.so timeout -k 10 400 python tools/exp_geom_tiles.py > $O/exp_geom_tiles.log 2>&1; echo "tiles rc=$?"; grep -v amdgpu.ids $O/exp_geom_tiles.log
timeout -k 10 200 python tools/stream_visibility_probe.py --reps 4 > $O/probe.log 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $O/probe.log
echo done
