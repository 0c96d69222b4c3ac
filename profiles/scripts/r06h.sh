#!/bin/bash
# round 6, VERDICT r5 item 6: (a) fp32 planned stiffness at P = 7, 8 -- the three builds (0 own buffers, 1 LDS-aliased + whole G slab,
# 2 LDS-aliased + G ring) and the two list encodings, interleaved; (b) config-2-size launches: the kernel back to back / HBM-cold /
# isolated against the host-timed single launch of the reference's protocol (tools/small_mesh_probe.py)
O=gpurun_out/r06h
mkdir -p $O
for cfg in "7 31" "8 27"; do
  set -- $cfg
  timeout -k 10 300 python tools/ab_stiffness.py --degree $1 --cells $2 --dtype f32 --rounds 7 --reps 20 plan plan:0 plan:1 plan:2 raw:1 raw:2 runs:2 2>&1 | grep -v Warning
done | tee $O/ab_fp32_p78.log
timeout -k 10 300 python tools/small_mesh_probe.py 20,25,29,32 2>&1 | grep -v Warning | tee $O/small_mesh.log
