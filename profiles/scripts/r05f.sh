#!/bin/bash
# round 5: chained plans for the in-kernel-geometry kernel (one workgroup per chain of sideways-neighbour batches, shared face carried in LDS)
O=gpurun_out/r05f
mkdir -p $O
timeout -k 10 500 python tools/exp_geom_chain.py > $O/exp_geom_chain.log 2>&1; echo "chain rc=$?"; grep -v amdgpu.ids $O/exp_geom_chain.log | tail -20
echo done
