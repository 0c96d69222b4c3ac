#!/bin/bash
# r04c: fork signal attached to the interior launch: ABI / adaptor tests, then the paired proxy A/B (attach off / on)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04c
timeout -k 10 1000 python -m pytest tests/test_abi.py tests/test_dolfinx_adaptor.py tests/test_bench_launch.py -m gpu -x -q > gpurun_out/r04c/tests.log 2>&1
rc=$?
tail -5 gpurun_out/r04c/tests.log
[ $rc -ne 0 ] && exit $rc
for i in 1 2; do
  for att in 0 1; do
    echo "== FUS_HALO_ATTACH_SYNC=$att (round $i)" >> gpurun_out/r04c/paired.log
    FUS_HALO_ATTACH_SYNC=$att timeout -k 10 300 python tools/overlap_probe.py --transport peer --paired 7 --reps 40 >> gpurun_out/r04c/paired.log 2>&1 || exit 1
  done
done
grep -E "^==|^paired" gpurun_out/r04c/paired.log
