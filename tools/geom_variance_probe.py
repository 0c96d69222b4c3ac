#!/usr/bin/env python3
"""Why does the in-kernel-geometry stiffness kernel read 0.155 ms in a fresh process (tools/exp_geom_tiles.py) and 0.178-0.185 ms
in the default bench line's ``aux`` (same mesh, same operator)?  One process, config 3; the kernel is timed (K back-to-back launches
between one event pair, like bench.py's aux) at several points of a bench-like sequence, with the device clocks from sysfs."""
import argparse
import glob
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def clocks():
    out = {}
    for card in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk"):
            try:
                with open(os.path.join(card, name)) as f:
                    cur = [ln.split(":", 1)[1].strip().rstrip("*").strip() for ln in f.read().splitlines() if ln.rstrip().endswith("*")]
                if cur:
                    out[name[7:]] = cur[0]
            except OSError:
                pass
        if out:
            break
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--K", type=int, default=20)
    a = ap.parse_args()
    import torch

    import fusgpu_loader

    ops, gll, boxmesh, pre = (fusgpu_loader.submodule(m) for m in ("operators", "gll", "boxmesh", "precompute"))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    P, N = 4, 54
    n = P + 1
    mesh = boxmesh.BoxMesh(P, N, perturb=0.16, seed=0)
    pts, wts, D = gll.tabulate_1d(P)
    xyz = mesh.dof_coordinates()
    x = torch.from_numpy(100 * np.sin(2 * np.pi * xyz[:, 0]) * np.cos(3 * np.pi * xyz[:, 1]) * np.sin(4 * np.pi * xyz[:, 2])).to(dev)
    y = torch.zeros_like(x)
    cc = torch.from_numpy(np.random.default_rng(1234).standard_normal(mesh.ncells)).to(dev)
    dm = torch.from_numpy(mesh.dofmap).to(dev)
    gop = ops.stiffness_operator(P, D.flatten(), np.float64, geometry=(mesh.x_dofs, mesh.x_g, pts, wts))
    gop.prepare(dm)

    def time_kernel(fn, K):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / K * 1e3

    def geom(label):
        ts = [time_kernel(lambda: gop(x, cc, y, None, dm), a.K) for _ in range(3)]
        long = time_kernel(lambda: gop(x, cc, y, None, dm), 500)
        print(f"{label:58s} geom kernel: K={a.K}: {ts[0]:6.1f} {ts[1]:6.1f} {ts[2]:6.1f} us | K=500: {long:6.1f} us | clocks {clocks()}", flush=True)

    geom("fresh process, nothing else allocated")
    G = torch.empty((mesh.ncells, n**3, 6), dtype=torch.float64, device=dev)
    w3 = gll.tensor_weights_3d(wts)
    dg = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts), np.float64)
    pre.compute_scaled_geometrical_factor_device(G, (torch.from_numpy(mesh.x_dofs).to(dev), torch.from_numpy(mesh.x_g).to(dev)), mesh.ncells,
                                                 torch.from_numpy(dg).to(dev), torch.from_numpy(w3).to(dev))
    op = ops.stiffness_operator(P, D.flatten(), np.float64)
    op.prepare(dm)
    geom("after allocating G (945 MB) and the row plan")
    print(f"{'':58s} headline kernel K=20: {time_kernel(lambda: op(x, cc, y, G, dm), 20):6.1f} us", flush=True)
    geom("after 25 headline applies")
    t = time_kernel(lambda: op(x, cc, y, G, dm), 2500)
    print(f"{'':58s} headline kernel K=2500: {t:6.1f} us", flush=True)
    geom("right after 2500 headline applies (0.56 s of load)")
    time.sleep(2.0)
    geom("after 2 s of idle")
    big = torch.empty(1 << 27, dtype=torch.float64, device=dev)
    for _ in range(50):
        big.fill_(1.0)
    torch.cuda.synchronize()
    geom("after 50 fills of 1 GiB (dirty lines in the memory-side cache)")
    y2 = torch.zeros_like(y)
    ts = time_kernel(lambda: gop(x, cc, y2, None, dm), 500)
    print(f"{'a fresh output vector':58s} geom kernel K=500: {ts:6.1f} us", flush=True)
    ops.use_strip_order(True)  # opt-in (default: rows): two adjacent rows interleaved, 2 x 5 pieces per batch
    ts = time_kernel(lambda: gop(x, cc, y, None, dm), 500)
    print(f"{'two-row strips instead of rows':58s} geom kernel K=500: {ts:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
