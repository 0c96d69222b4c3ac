#!/bin/bash
# mass gather kernel after the row-length fix: GPU tests that touch the mass operators, then its rocprofv3 passes (default / static detJ)
set -e
O=gpurun_out/r05z
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "mass or gather or solver or abi or golden" > $O/pytest_mass.log 2>&1 || { tail -20 $O/pytest_mass.log; exit 1; }
tail -2 $O/pytest_mass.log
prof() { tag=$1; shift; bash profiles/run_profile.sh $tag "$@" > $O/prof_$tag.log 2>&1 || { tail -20 $O/prof_$tag.log; exit 1; }; echo "$tag done"; }
prof r05z_mass --mode mass
prof r05z_mass_static --mode mass --mass-static
echo profiles F done
