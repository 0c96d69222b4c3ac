#!/bin/bash
# after the last kernel-source change of round 6 (in-loop flux form at fp64 P = 2, 3 / fp32 P = 2, 3, 4; a comment in rk4.hpp): GPU tests of everything
# that forms geometry or steps in time, then every rocprofv3 pass whose kernel sources changed (same tags: they replace the earlier ones)
set -e
O=gpurun_out/r06z
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "geom or geometry or westervelt or solver or rk4 or config5 or golden or degrees" > $O/pytest_geom.log 2>&1 || { tail -20 $O/pytest_geom.log; exit 1; }
tail -2 $O/pytest_geom.log
prof() { tag=$1; shift; bash profiles/run_profile.sh $tag "$@" > $O/prof_$tag.log 2>&1 || { tail -20 $O/prof_$tag.log; exit 1; }; echo "$tag done"; }
prof r06z_geom --mode stiffness_geom
prof r06z_rk4 --mode rk4 --perturbed
prof r06z_rk4_geom --mode rk4 --perturbed --in-kernel-geometry
prof r06z_westervelt --mode westervelt --degree 6 --cells 36
prof r06z_westervelt_geom --mode westervelt --degree 6 --cells 36 --in-kernel-geometry
prof r06z_westervelt_geom_single_gather --mode westervelt --degree 6 --cells 36 --in-kernel-geometry --single-gather
( timeout -k 10 600 python tools/sweep.py --degrees 2,3,4 2>&1 | grep "^P=" ) | tee $O/sweep_p234.log | cut -c1-250
echo profiles r06z_c done
