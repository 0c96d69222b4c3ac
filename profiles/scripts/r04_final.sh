#!/bin/bash
# rocprofv3 passes of the round-4 library: headline kernel, mass kernel (shipped / with exclusive-dof marks: TCC_EA0_ATOMIC
# before / after), the fused RK4 step with the general G and with in-kernel geometry (per-kernel traffic -> aux.rk4_step*)
set -e
O=gpurun_out/r04_final
mkdir -p $O
bash profiles/run_profile.sh r04_final > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
bash profiles/run_profile.sh r04_final_mass --mode mass > $O/prof_mass.log 2>&1 || { tail -20 $O/prof_mass.log; exit 1; }
bash profiles/run_profile.sh r04_final_mass_exclusive --mode mass --exclusive > $O/prof_mass_excl.log 2>&1 || { tail -20 $O/prof_mass_excl.log; exit 1; }
bash profiles/run_profile.sh r04_final_rk4 --mode rk4 --perturbed > $O/prof_rk4.log 2>&1 || { tail -20 $O/prof_rk4.log; exit 1; }
bash profiles/run_profile.sh r04_final_rk4_geom --mode rk4 --perturbed --in-kernel-geometry > $O/prof_rk4_geom.log 2>&1 || { tail -20 $O/prof_rk4_geom.log; exit 1; }
echo profiles done
