#!/bin/bash
# A/B of two builds of the library on one box: base (round-3 kernel prologue) vs new (loads issued without intervening uses)
set -e
O=gpurun_out/r03w
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_operators_gpu.py -x -q -m gpu -k "golden or full_size_config3 or config2" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
BASE=$(pwd)/fenicsx-fus-gpu_amd/csrc/_ab/libfusgpu_base.so
{
for i in 1 2 3; do
  echo "== base";  FUS_LIB_PATH=$BASE timeout -k 10 200 python tools/ab_stiffness.py --rounds 5 --reps 20 plan 2>&1 | tail -3
  echo "== new";   timeout -k 10 200 python tools/ab_stiffness.py --rounds 5 --reps 20 plan 2>&1 | tail -3
done
} > $O/ab.log 2>&1 || { tail -30 $O/ab.log; exit 1; }
cat $O/ab.log
