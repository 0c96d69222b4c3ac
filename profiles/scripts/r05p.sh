#!/bin/bash
# round 5: what bounds the in-kernel-geometry kernel -- phase clocks (instrumented build of a COPY of csrc/, -DFUS_ABLATE=32: s_memtime /
# s_memrealtime at 6 points, thread 0 of every workgroup), the same with the geometry arithmetic removed (36), and the IEEE division of
# column_g_at replaced by v_rcp_f64 without (8) and with two Newton steps (16).  The shipped sources are untouched.
O=gpurun_out/r05p
mkdir -p $O
for a in "" 32 64 96 ""; do
  if [ -z "$a" ]; then lib=$PWD/fenicsx-fus-gpu_amd/csrc/libfusgpu.so; else lib=$PWD/tools/_bin/libfusgpu_ablate$a.so; fi
  FUS_LIB_PATH=$lib timeout -k 10 200 python tools/ablate_geom.py 2>&1 | grep -E "^P=|^   " | sed "s/^/ablate bits ${a:-0}: /"
done | tee $O/ablate_geom_phases.log
