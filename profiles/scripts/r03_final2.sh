#!/bin/bash
# final state of round 3 (arenas in uncached memory): halo + bench-launch GPU tests, the judged halo comparison, then the
# rocprofv3 passes of THIS library for the headline kernel and the mass kernel
set -e
O=gpurun_out/r03_final2
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_halo_gpu.py tests/test_bench_launch.py tests/test_abi.py tests/test_rk4_golden.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 300 python tools/overlap_probe.py --transport ipc --paired 9 --reps 40 2>&1 | grep "^paired" | tee $O/paired.log
bash profiles/run_profile.sh r03_final > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
bash profiles/run_profile.sh r03_final_mass --mode mass > $O/prof_mass.log 2>&1 || { tail -20 $O/prof_mass.log; exit 1; }
echo profiles done
