#!/bin/bash
# round 5: single-gather vs two-gather Westervelt cell pass, with the G array and with in-kernel geometry (paired); then the profile passes, part A
O=gpurun_out/r05i
mkdir -p $O
timeout -k 10 400 python tools/ab_westervelt_gathers.py > $O/ab_westervelt_gathers.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids $O/ab_westervelt_gathers.log | tail -6
bash profiles/scripts/r05_final_a.sh
