#!/bin/bash
# rocprofv3 passes of the round-5 library after the preamble restructure, part B: the fused RK4 step (general G / in-kernel geometry) and the Westervelt P = 6 step
# (general G; in-kernel geometry = the solver's default, two-gather cell pass; its single-gather form)
set -e
O=gpurun_out/r05z
mkdir -p $O
prof() { tag=$1; shift; bash profiles/run_profile.sh $tag "$@" > $O/prof_$tag.log 2>&1 || { tail -20 $O/prof_$tag.log; exit 1; }; echo "$tag done"; }
prof r05z_rk4 --mode rk4 --perturbed
prof r05z_rk4_geom --mode rk4 --perturbed --in-kernel-geometry
prof r05z_westervelt --mode westervelt --degree 6 --cells 36
prof r05z_westervelt_geom --mode westervelt --degree 6 --cells 36 --in-kernel-geometry
prof r05z_westervelt_geom_single_gather --mode westervelt --degree 6 --cells 36 --in-kernel-geometry --single-gather
echo profiles B done
