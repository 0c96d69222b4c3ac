#!/bin/bash
# rocprofv3 passes of the FINAL round-6 library (plan.hpp, rk4.hpp, westervelt.hpp, halo_ipc.hpp changed since r05z): headline kernel, mass
# (default gather / static detJ), in-kernel geometry, the fused RK4 step (G array / in-kernel geometry; LEAN stage kinds) and the Westervelt
# P = 6 step (G array / in-kernel geometry / its single-gather form).  Condensed here by profiles/scripts/r06z_summarize.sh.
set -e
O=gpurun_out/r06z
mkdir -p $O
prof() { tag=$1; shift; bash profiles/run_profile.sh $tag "$@" > $O/prof_$tag.log 2>&1 || { tail -20 $O/prof_$tag.log; exit 1; }; echo "$tag done"; }
prof r06z
prof r06z_mass --mode mass
prof r06z_mass_static --mode mass --mass-static
prof r06z_geom --mode stiffness_geom
prof r06z_rk4 --mode rk4 --perturbed
prof r06z_rk4_geom --mode rk4 --perturbed --in-kernel-geometry
prof r06z_westervelt --mode westervelt --degree 6 --cells 36
prof r06z_westervelt_geom --mode westervelt --degree 6 --cells 36 --in-kernel-geometry
prof r06z_westervelt_geom_single_gather --mode westervelt --degree 6 --cells 36 --in-kernel-geometry --single-gather
echo profiles r06z done
