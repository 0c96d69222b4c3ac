#!/usr/bin/env python3
"""How does a SHORT timed region (K = 20 launches between two synchronisations, what the driver times) compare with the
steady state of the headline kernel?  Regions of 20 launches after different preludes (warm-up count, idle gap), each
split into windows of 5 launches by HIP events; then 500 back-to-back launches."""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import fusgpu_loader  # noqa: E402
from conftest import build_problem  # noqa: E402

ops = fusgpu_loader.submodule("operators")
pb = build_problem(4, 54, perturb=0.16)
mesh = pb["mesh"]
dev = torch.device("cuda", 0)
x, cc, G = (torch.from_numpy(pb[k]).to(dev) for k in ("x", "cc", "G"))
dm = torch.from_numpy(mesh.dofmap).to(dev)
y = torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev)
op = ops.stiffness_operator(4, pb["D"].flatten(), np.float64)
op.prepare(dm)


def step():
    op(x, cc, y, G, dm)


def region(K=20, win=5):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K // win + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for w in range(K // win):
        for _ in range(win):
            step()
        ev[w + 1].record()
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) / win * 1e3 for i in range(K // win)], ev[0].elapsed_time(ev[-1]) / K * 1e3


def region_plain(K=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(K):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K * 1e3


for name, warm, gap in (("5 warm-up launches, no gap", 5, 0.0), ("5 warm-up launches, no gap (again)", 5, 0.0), ("100 warm-up launches", 100, 0.0),
                        ("5 warm-up launches, 5 ms idle", 5, 0.005), ("5 warm-up launches, 200 ms idle", 5, 0.2), ("no warm-up after 1 s idle", 0, 1.0)):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    if gap:
        time.sleep(gap)
    plain = region_plain()
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    if gap:
        time.sleep(gap)
    wins, avg = region()
    print(f"{name:40s} 20-launch region {plain:7.1f} us/launch | with window events {avg:7.1f}: windows of 5: " + " ".join(f"{w:6.1f}" for w in wins), flush=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(500):
    step()
e1.record()
torch.cuda.synchronize()
print(f"500 back-to-back launches: {e0.elapsed_time(e1) / 500 * 1e3:.1f} us/launch")
for K in (20, 40, 100, 200):
    print(f"region of {K}: {region_plain(K):.1f} us/launch", flush=True)
