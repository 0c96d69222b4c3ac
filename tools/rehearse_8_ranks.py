#!/usr/bin/env python3
"""BASELINE config 4 at FULL SIZE with real processes, as far as a one-GPU box allows: the 8-rank 2x2x2 partition of a P = 4,
108^3-cell box (81.2 M dofs; per rank 54^3 cells, 3 face messages of 47 089 dofs, 3 edges of 217, 1 corner -- the message
sizes of the 8-GPU run) on 4 PROCESSES of 2 ranks each sharing the GPU (the pool's process guard admits at most 6 processes
on the card).  Neighbours of the same process exchange through plain pointers, the others through HIP IPC mappings -- the
PEER transport's own kernels either way.  Checks, like bench.py's own halo check: poisoned ghosts come back from the forward
exchange as the analytic field exactly; the sum of y over the owned dofs of all ranks equals 1^T K x = 0; no device-side
wait failed.  Then a few applies are timed (NOT a measurement: 8 ranks share one GPU).

    python tools/rehearse_8_ranks.py [--cells 54] [--degree 4] [--applies 5]        # parent: starts the 4 processes
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def worker(a):
    proc, nproc = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("FUS_IPC_SPIN_SECONDS", "120")  # 8 ranks' queues time-sliced on one card
    import torch
    import torch.distributed as dist

    import fusgpu_loader

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=proc, world_size=nproc)
    scat, boxmesh, utils, ops, gll, pre = (fusgpu_loader.submodule(m) for m in ("scatterer", "boxmesh", "utils", "operators", "gll", "precompute"))
    P, grid, R, rpp = a.degree, (2, 2, 2), 8, 2
    cells = (2 * a.cells,) * 3
    mine = list(range(proc * rpp, (proc + 1) * rpp))
    t0 = time.time()
    imaps, meshes = [], {}
    for r in range(R):  # every rank's index map (the halo plan needs all of them); only this process's meshes are kept
        m = boxmesh.BoxMesh(P, cells, grid=grid, rank=r, perturb=0.16, seed=0)
        imaps.append(m.index_map)
        if r in mine:
            meshes[r] = m
        del m
    od, gd = utils.compute_scatterer_data_all(imaps)
    n = P + 1
    pts, wts, D = gll.tabulate_1d(P)
    op = ops.stiffness_operator(P, D.flatten(), np.float64)
    dphi = torch.from_numpy(pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))).to(dev)
    w3 = torch.from_numpy(gll.tensor_weights_3d(wts)).to(dev)
    ranks, halos = [], []
    for r in mine:
        mesh = meshes[r]
        xyz = mesh.dof_coordinates()
        x = 100 * np.sin(2 * np.pi * xyz[:, 0]) * np.cos(3 * np.pi * xyz[:, 1]) * np.sin(4 * np.pi * xyz[:, 2])
        del xyz
        G = torch.empty((mesh.ncells, n**3, 6), dtype=torch.float64, device=dev)
        pre.compute_scaled_geometrical_factor_device(G, (torch.from_numpy(mesh.x_dofs).to(dev), torch.from_numpy(mesh.x_g).to(dev)), mesh.ncells, dphi, w3)
        rk = dict(mesh=mesh, x=torch.from_numpy(x).to(dev), y=torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev),
                  cc=torch.from_numpy(np.random.default_rng(1234 + r).standard_normal(mesh.ncells)).to(dev), G=G,
                  dm=torch.from_numpy(mesh.dofmap).to(dev))
        comm = scat.NativeComm(local=(8800, R, r), transport="peer", hosted=mine)
        ranks.append(rk)
        halos.append(scat.HaloApply(mesh, op, comm, np.float64, plan=(od[r], gd[r]), schedule="concurrent"))
    setup_s = time.time() - t0

    def lockstep(gens):
        live = list(gens)
        while live:
            nxt = []
            for g in live:
                try:
                    next(g)
                    nxt.append(g)
                except StopIteration:
                    pass
            live = nxt

    dist.barrier()
    # forward: poisoned ghosts must come back as the analytic field, exactly
    expect = []
    for rk in ranks:
        nl = rk["mesh"].nlocal
        expect.append(rk["x"][nl:].clone())
        rk["x"][nl:] = -777.0
    for h, rk in zip(halos, ranks):
        h.fwd.begin(rk["x"])
    for h, rk in zip(halos, ranks):
        h.fwd.end(rk["x"])
    torch.cuda.synchronize()
    fwd_err = max(float((rk["x"][rk["mesh"].nlocal:] - e).abs().max().item()) if e.numel() else 0.0 for rk, e in zip(ranks, expect))
    # the whole apply (folded fork / join; the fork's signal on the interior launch from the second apply on)
    t_apply = []
    for rep in range(a.applies):
        for rk in ranks:
            rk["y"].zero_()
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        lockstep([h.apply_schedule(rk["x"], rk["cc"], rk["y"], rk["G"], rk["dm"]) for h, rk in zip(halos, ranks)])
        torch.cuda.synchronize()
        t_apply.append(time.perf_counter() - t1)
    sums = torch.zeros(4, dtype=torch.float64)
    for h, rk in zip(halos, ranks):
        nl = rk["mesh"].nlocal
        sums += torch.tensor([float(rk["y"][:nl].sum().item()), float(rk["y"][:nl].abs().sum().item()), float(h.health()), fwd_err], dtype=torch.float64)
    dist.all_reduce(sums)
    rel = abs(float(sums[0])) / max(float(sums[1]), 1e-300)
    ok = rel < 1e-9 and float(sums[2]) == 0.0 and float(sums[3]) == 0.0
    if proc == 0:
        m0 = ranks[0]["mesh"]
        nb = {int(r_): int(c_) for r_, c_ in zip(gd[0][3], gd[0][1])}
        print(json.dumps({"what": "config 4 at full size: 8 ranks (2x2x2) on 4 processes x 2 ranks sharing ONE GPU, PEER transport (rehearsal, NOT a measurement)",
                          "degree": P, "cells_per_rank": m0.ncells, "global_dofs": m0.ndofs_global, "ranks": R, "processes": nproc,
                          "rank0_ghosted_by": nb, "forward_max_abs_err": float(sums[3]), "owned_sum_defect_over_sum_abs": rel,
                          "failed_device_waits": int(sums[2]), "ok": bool(ok), "schedule": halos[0].schedule_kind, "setup_s": setup_s,
                          "apply_s_per_rep": t_apply}), flush=True)
    dist.barrier()
    for h in halos:
        h.fwd.close(), h.rev.close()
    dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cells", type=int, default=54, help="cells per direction PER RANK")
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--applies", type=int, default=5)
    a = ap.parse_args()
    if "RANK" in os.environ:
        raise SystemExit(worker(a))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(4):  # the parent never touches the GPU
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    raise SystemExit(rc)


if __name__ == "__main__":
    main()
