#!/bin/bash
# rocprofv3 passes of the round-5 library after the preamble restructure, part A: headline kernel, mass (default gather kernel / static-detJ form), in-kernel geometry
set -e
O=gpurun_out/r05z
mkdir -p $O
prof() { tag=$1; shift; bash profiles/run_profile.sh $tag "$@" > $O/prof_$tag.log 2>&1 || { tail -20 $O/prof_$tag.log; exit 1; }; echo "$tag done"; }
prof r05z
prof r05z_mass --mode mass
prof r05z_mass_static --mode mass --mass-static
prof r05z_geom --mode stiffness_geom
echo profiles A done
