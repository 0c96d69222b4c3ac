#!/bin/bash
# round 5: numbering experiment after the plan-level run-table decision (a launch reads the run tables only if at least half of the
# plan's batches carry one), then the GPU suite
O=gpurun_out/r05y
mkdir -p $O
timeout -k 10 500 python tools/exp_numbering.py 2>&1 | grep -v "^\[\|amdgpu.ids" | tee $O/numbering_plan_level_runs.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; tail -3 $O/pytest_gpu.log; exit $rc
