#!/usr/bin/env python3
"""Why does the operator run 15 % slower inside the RK4 loop than back to back?  The headline apply timed with one event pair
per launch, preceded by different kernels on the same stream: nothing, a busy-wait that touches no memory, a read-only /
write-only / copy stream of 1 GiB, the fused vector pass itself (10 vectors of the mesh's size)."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import fusgpu_loader  # noqa: E402
from conftest import build_problem  # noqa: E402

ops, _lib = fusgpu_loader.submodule("operators"), fusgpu_loader.submodule("_lib")
lib = _lib.load()
pb = build_problem(4, 54, perturb=0.16)
mesh = pb["mesh"]
dev = torch.device("cuda", 0)
x, cc, G = (torch.from_numpy(pb[k]).to(dev) for k in ("x", "cc", "G"))
dm = torch.from_numpy(mesh.dofmap).to(dev)
y = torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev)
op = ops.stiffness_operator(4, pb["D"].flatten(), np.float64)
op.prepare(dm)
big = torch.ones((1 << 30) // 8, dtype=torch.float64, device=dev)
big2 = torch.empty_like(big)
vecs = [torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev) for _ in range(8)]
minv = torch.ones(mesh.ndofs, dtype=torch.float64, device=dev)


def vector_pass(kind):
    b, u, v, u0, v0, ku, un = vecs[:7]
    _lib.check(lib.fus_rk4_stage_f64(1e-9, 1e-9, kind, minv.data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), u0.data_ptr(), v0.data_ptr(),
                                     ku.data_ptr(), un.data_ptr(), mesh.ndofs, mesh.ndofs, _lib.stream_ptr()), "rk4")


preludes = {
    "nothing (isolated launches)": lambda: None,
    "busy wait ~150 us, no memory traffic": lambda: torch.cuda._sleep(360000),
    "read-only stream of 1 GiB (torch.sum)": lambda: big.sum(),
    "write-only stream of 1 GiB (fill)": lambda: ops.fill(1.0, big2),
    "copy 1 GiB -> 1 GiB": lambda: ops.copy(big, big2),
    "the vector pass, MIDDLE stage (12 touches)": lambda: vector_pass(0),
    "the vector pass, LAST stage (8 touches)": lambda: vector_pass(3),
    "the vector pass, then a 150 us busy wait": lambda: (vector_pass(0), torch.cuda._sleep(360000)),
}
for rnd in range(2):
    for name, pre in preludes.items():
        ts = []
        for _ in range(30):
            pre()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            op(x, cc, y, G, dm)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = np.array(ts[5:])
        print(f"round {rnd}: stiffness apply after {name:45s} {np.median(ts):7.1f} us (min {ts.min():6.1f}, max {ts.max():6.1f})", flush=True)
