#!/bin/bash
# round 5: chained plans, last two questions: the chain ORDER alone (chain length -1: same single-batch kernel), and the chained kernel built
# WITHOUT the occupancy hint (163 VGPRs, 3 workgroups per CU, no spills) -- FUS_LIB_PATH variant
O=gpurun_out/r05g
mkdir -p $O
timeout -k 10 400 python tools/exp_geom_chain.py --chains 1,-1,2,4 --cases 4:54 > $O/exp_geom_chain_hinted.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/exp_geom_chain_hinted.log | tail -6
FUS_LIB_PATH=$PWD/tools/_bin/libfusgpu_chain_nohint.so timeout -k 10 400 python tools/exp_geom_chain.py --chains 1,2,4 --cases 4:54,6:36 > $O/exp_geom_chain_nohint.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $O/exp_geom_chain_nohint.log | tail -8
echo done
