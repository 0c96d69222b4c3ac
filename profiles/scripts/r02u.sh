python tools/exp_batch_shape.py 2>&1 | grep -v amdgpu
