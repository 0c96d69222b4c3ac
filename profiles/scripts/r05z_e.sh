#!/bin/bash
# rocprofv3 passes of the kernels that form the geometry themselves, after the reciprocal replaced the division in column_g_at;
# GPU tests that touch in-kernel geometry first
set -e
O=gpurun_out/r05z
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "geom or geometry or westervelt or solver or rk4" > $O/pytest_geom.log 2>&1 || { tail -20 $O/pytest_geom.log; exit 1; }
tail -2 $O/pytest_geom.log
prof() { tag=$1; shift; bash profiles/run_profile.sh $tag "$@" > $O/prof_$tag.log 2>&1 || { tail -20 $O/prof_$tag.log; exit 1; }; echo "$tag done"; }
prof r05z_geom --mode stiffness_geom
prof r05z_rk4_geom --mode rk4 --perturbed --in-kernel-geometry
prof r05z_westervelt_geom --mode westervelt --degree 6 --cells 36 --in-kernel-geometry
prof r05z_westervelt_geom_single_gather --mode westervelt --degree 6 --cells 36 --in-kernel-geometry --single-gather
echo profiles E done
