#!/bin/bash
# round 5, final library: the reference's numbering experiment (tools/exp_numbering.py; profiles/r02d_numbering_locality_plan.log had the round-2 kernels)
O=gpurun_out/r05y
mkdir -p $O
timeout -k 10 500 python tools/exp_numbering.py 2>&1 | grep -v "^\[\|amdgpu.ids" | tee $O/numbering.log
