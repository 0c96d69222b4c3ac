#!/bin/bash
# round 5: single- vs two-gather Westervelt pass at config 3's shape (P = 4, 54^3); then the profile passes, part B
O=gpurun_out/r05j
mkdir -p $O
timeout -k 10 300 python tools/ab_westervelt_gathers.py --degree 4 --cells 54 --rounds 3 > $O/ab_westervelt_gathers_P4.log 2>&1; echo "ab rc=$?"; grep -v amdgpu.ids $O/ab_westervelt_gathers_P4.log | tail -5
bash profiles/scripts/r05_final_b.sh
