#!/bin/bash
# the geometry-forming kernels changed after r06z.sh (column_flux_at at fp64 P = 6, 8, 9 and in the Westervelt P = 6 cell pass): their GPU tests, then
# their rocprofv3 passes again (same tags: they replace the first ones)
set -e
O=gpurun_out/r06z
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "geom or geometry or westervelt or solver or rk4 or config5 or golden" > $O/pytest_geom.log 2>&1 || { tail -20 $O/pytest_geom.log; exit 1; }
tail -2 $O/pytest_geom.log
prof() { tag=$1; shift; bash profiles/run_profile.sh $tag "$@" > $O/prof_$tag.log 2>&1 || { tail -20 $O/prof_$tag.log; exit 1; }; echo "$tag done"; }
prof r06z_geom --mode stiffness_geom
prof r06z_rk4_geom --mode rk4 --perturbed --in-kernel-geometry
prof r06z_westervelt_geom --mode westervelt --degree 6 --cells 36 --in-kernel-geometry
prof r06z_westervelt_geom_single_gather --mode westervelt --degree 6 --cells 36 --in-kernel-geometry --single-gather
echo profiles r06z_b done
