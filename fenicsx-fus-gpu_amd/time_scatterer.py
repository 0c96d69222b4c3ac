#!/usr/bin/env python3
"""
Scatterer timing harness -- the counterpart of the reference's numba-cpu/time_scatterer.py (:126-210: 50 timed
``scatter_rev(u)`` and 10 timed ``scatter_fwd(u)`` calls, one call at a time between two clock reads, mean +/- std) on
the MI355X closures.  The reference partitions a 4^3-cell P = 4 box over its MPI ranks; here the box is the synthetic
structured one, split over the ranks of ``torch.distributed`` (one process per GPU), and ``--cells`` is a parameter.

    python -m torch.distributed.run --nproc-per-node 8 fenicsx-fus-gpu_amd/time_scatterer.py --cells 108
    python fenicsx-fus-gpu_amd/time_scatterer.py --self-neighbour      # one GPU: a rank that is its own neighbour, config-4 messages

The closures do not block the host (the reference's do: MPI Waitall), so each timed call is followed by a device
synchronise, as ``bench.py --mode scatter`` does; FUS_HALO picks the transport (peer | native | torch).
"""

import argparse
import os
import sys
from time import perf_counter_ns

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--cells", type=int, default=4, help="cells per direction of the whole box (the reference: 4)")
    ap.add_argument("--self-neighbour", action="store_true", help="one rank that is its own neighbour with BASELINE config-4 message sizes")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    import fusgpu_loader

    boxmesh, scat, utils = (fusgpu_loader.submodule(m) for m in ("boxmesh", "scatterer", "utils"))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    dev = torch.device("cuda", torch.cuda.current_device())
    ft = np.float64 if a.dtype == "f64" else np.float32
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
        comm = scat.default_comm()
        mesh = boxmesh.BoxMesh(a.degree, a.cells, grid=boxmesh.default_grid(world), rank=rank, dtype=ft)
        owners_data, ghosts_data = utils.compute_scatterer_data_flat(mesh.index_map, comm)
        nlocal, ndofs = mesh.nlocal, mesh.ndofs
        xyz = mesh.dof_coordinates()
        u_ = 100 * np.sin(2 * np.pi * xyz[:, 0]) * np.cos(3 * np.pi * xyz[:, 1]) * np.sin(4 * np.pi * xyz[:, 2])  # :108-116
    else:
        comm = scat.NativeComm(transport={"peer": "peer", "native": "rccl"}.get(os.environ.get("FUS_HALO", "peer"), "peer"))
        if a.self_neighbour:
            owners_data, ghosts_data, nlocal = utils.config4_self_plan(a.degree * 54 + 1)
            ndofs = nlocal + int(owners_data[1][0])
        else:  # one rank, no neighbours: the exchanges are no-ops (the reference on one MPI rank is, too)
            mesh = boxmesh.BoxMesh(a.degree, a.cells, dtype=ft)
            owners_data, ghosts_data = utils.compute_scatterer_data_flat(mesh.index_map, None)
            nlocal, ndofs = mesh.nlocal, mesh.ndofs
        u_ = np.random.default_rng(0).standard_normal(ndofs)
    u = torch.from_numpy(u_.astype(ft)).to(dev)
    for name, mk, reps in (("scatter reverse", scat.scatter_reverse, 50), ("scatter forward", scat.scatter_forward, 10)):
        sc = mk(comm, owners_data, ghosts_data, nlocal, ft)
        sc(u)  # the reference calls once to JIT-compile; here: arenas connected, code objects loaded
        torch.cuda.synchronize()
        t = np.empty(reps)
        for i in range(reps):
            tic = perf_counter_ns()
            sc(u)
            torch.cuda.synchronize()
            t[i] = perf_counter_ns() - tic
        t *= 1e-9
        if rank == 0:
            print(f"Elapsed time ({name} (MI355X)): {t.mean():.7f} ± {t.std():.7f} s", flush=True)
        if hasattr(sc, "status") and sc.status().get("failures", 0):
            raise SystemExit(f"rank {rank}: {name}: device-side waits of the exchange failed")
        if hasattr(sc, "close"):
            sc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
