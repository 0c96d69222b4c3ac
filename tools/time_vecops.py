#!/usr/bin/env python3
"""Bandwidth of the streaming vector kernels and of the fused RK4 stage kernel at 10.2 M dofs (fp64)."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import fusgpu_loader  # noqa: E402

ops, lib = fusgpu_loader.submodule("operators"), fusgpu_loader.submodule("_lib")
torch.cuda.set_device(0)
n = 10218313
a, b, c = (torch.rand(n, dtype=torch.float64, device="cuda") + 1 for _ in range(3))
vecs = [torch.rand(n, dtype=torch.float64, device="cuda") + 1 for _ in range(8)]


def timeit(name, fn, nbytes, reps=50):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    print(f"{name:28s} {t * 1e3:8.1f} us  {nbytes / t / 1e6:7.0f} GB/s  ({100 * nbytes / t / 1e6 / 8000:4.1f}% of 8 TB/s)")


timeit("axpy (24 B/dof)", lambda: ops.axpy[1, 1](0.5, a, b), 24 * n)
timeit("copy (16 B/dof)", lambda: ops.copy(a, b), 16 * n)
timeit("fill (8 B/dof)", lambda: ops.fill(1.0, b), 8 * n)
timeit("pointwise_divide (24 B/dof)", lambda: ops.pointwise_divide(a, b, c), 24 * n)
timeit("square (16 B/dof)", lambda: ops.square(a, c), 16 * n)
fn = lib.load().fus_rk4_stage_f64
p = [v.data_ptr() for v in vecs]
timeit("fused rk4 stage (96 B/dof)", lambda: lib.check(fn(1e-9, 1e-9, 0, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], n, n, lib.stream_ptr())), 96 * n)
timeit("fused rk4 stage, new step (104)", lambda: lib.check(fn(1e-9, 0.0, 1, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], n, n, lib.stream_ptr())), 104 * n)
