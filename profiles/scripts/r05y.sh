#!/bin/bash
# round 5, final library: the reference's time_operators.py protocol (one apply per timing, 10 reps) at the sizes of profiles/r04v_time_operators.log,
# and the demo_linear_box time loop at config-3 size (fused path: affine box and perturbed cells)
O=gpurun_out/r05y
mkdir -p $O
{
for spec in "2 18" "4 25" "4 32" "4 54"; do
  set -- $spec
  echo "# time_operators.py --degree $1 --cells $2"
  timeout -k 10 200 python -c "import fusgpu_loader, sys; sys.argv = ['time_operators.py', '--degree', '$1', '--cells', '$2']; fusgpu_loader.submodule('time_operators').main()" 2>&1 | grep -v "^\["
done
} | tee $O/time_operators.log
