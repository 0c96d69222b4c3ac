#!/bin/bash
# r04a: halo + bench-launch tests with the poisoned flags / folded fork-join, then the paired proxy A/B
# (fork / join as kernels of their own vs folded into the first send / last receive kernel)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout -k 10 1000 python -m pytest tests/test_halo_gpu.py tests/test_bench_launch.py tests/test_solver_gpu.py tests/test_rk4_golden.py -m gpu -x -q > gpurun_out/r04a/halo_tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04a/halo_tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  for fold in 0 1; do
    echo "== FUS_HALO_FOLD_SYNC=$fold (round $i)" >> gpurun_out/r04a/paired.log
    FUS_HALO_FOLD_SYNC=$fold timeout -k 10 300 python tools/overlap_probe.py --transport peer --paired 7 --reps 40 >> gpurun_out/r04a/paired.log 2>&1 || exit 1
  done
done
grep -E "^==|^paired" gpurun_out/r04a/paired.log
