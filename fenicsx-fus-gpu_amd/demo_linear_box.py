#!/usr/bin/env python3
"""
Linear plane wave in a box, explicit RK4, on the MI355X operators -- the counterpart of the
reference's cuda/demo_linear_box.py (same physical parameters :53-80, time-step rule :115-122,
stage sequence :487-566, final-plane sampling :128-141) on the synthetic structured mesh that
replaces dolfinx's ``create_box``.

    python fenicsx-fus-gpu_amd/demo_linear_box.py [--cells N] [--degree P] [--reference-sequence] [--out file.npz]
    python -m torch.distributed.run --nproc-per-node 8 fenicsx-fus-gpu_amd/demo_linear_box.py   # one rank per GPU

Prints the reference's progress / timing lines (cuda/demo_linear_box.py:125,183,569-581) and
optionally stores the pressure field sampled on the z = 0 plane of the dof grid.
"""

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--cells", type=int, default=None, help="cells per direction of the WHOLE box, split over the ranks' blocks -- the reference's fixed-size "
                         "box, strong scaling (default: 2 per wavelength, as the reference)")
    ap.add_argument("--reference-sequence", action="store_true", help="the reference's unfused launch sequence")
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--eval-out", default=None, metavar="DIR",
                    help="evaluate the final pressure field at the reference's 100 x 100 points of the z = 0 plane and append the rows "
                         "'x,y,value' to DIR/pressure_field_nproc<N>.txt, rank after rank (cuda/demo_linear_box.py:128-141,587-605)")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    import fusgpu_loader

    boxmesh, ls, scat = (fusgpu_loader.submodule(m) for m in ("boxmesh", "linear_solver", "scatterer"))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    comm = None
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        comm = scat.default_comm()  # exchange issued by libfusgpu.so (PEER transport; FUS_HALO=native: RCCL); torch.distributed only bootstraps it

    # cuda/demo_linear_box.py:53-80
    float_type = np.float64
    source_frequency, source_amplitude = 0.5e6, 60000.0
    speed_of_sound, density = 1500.0, 1000.0
    domain_length = 0.12
    wave_length = speed_of_sound / source_frequency
    num_element = a.cells if a.cells is not None else int(2 * domain_length / wave_length)
    grid = boxmesh.default_grid(world)
    mesh = boxmesh.BoxMesh(a.degree, num_element, grid=grid, rank=rank, length=domain_length, dtype=float_type)
    h = ls.time_step_parameters(mesh, a.degree, speed_of_sound, source_frequency, domain_length)
    if world > 1:
        hm = torch.tensor([h], dtype=torch.float64, device="cuda")
        dist.all_reduce(hm, op=dist.ReduceOp.MIN)  # comm.Allreduce(hmin, mesh_size, op=MPI.MIN), :108
        h = float(hm.item())
    dt, tf, nstep = ls.snap_time_step(h, a.degree, speed_of_sound, source_frequency, domain_length)
    if rank == 0:
        print(f"Number of steps: {nstep}", flush=True)
        print(f"Number of degrees-of-freedom: {mesh.ndofs_global}", flush=True)
    solver = ls.LinearSpectral3D(mesh, float_type, speed_of_sound, density, source_frequency, source_amplitude,
                                 comm=comm, fused=not a.reference_sequence)
    solver.init()
    if rank == 0:
        print("Solve!", flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t, steps = solver.rk4(0.0, tf, dt, max_steps=a.max_steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if rank == 0:
        print(f"t: {t:5.5},\t Steps: {steps}/{nstep}", flush=True)
        print(f"Solve time: {el}")
        print(f"Solve time per step: {el / max(steps, 1)}")
    if a.out:
        lex = mesh.global_lexicographic_ids()[: mesh.nlocal]
        gd = mesh.global_dof_dims
        on_plane = (lex % gd[2]) == 0  # z = 0 plane of the dof grid
        np.savez(a.out if world == 1 else f"{a.out}.rank{rank}", lex=lex[on_plane], u=solver.u_sol()[on_plane],
                 dims=np.array(gd), t=t, steps=steps)
    if a.eval_out:
        pe = fusgpu_loader.submodule("point_evaluation")
        xp = np.linspace(0, domain_length, 100, dtype=float_type)  # :130-139
        X_p, Y_p = np.meshgrid(xp, xp)
        points = np.zeros((3, 100 * 100), dtype=float_type)
        points[0], points[1] = X_p.flatten(), Y_p.flatten()
        x_eval, cell_eval = pe.compute_eval_params(mesh, points, float_type)
        u_full = solver.u_sol(with_ghosts=True)
        data = np.zeros_like(x_eval)
        if len(cell_eval):
            data[:, 0], data[:, 1] = x_eval[:, 0], x_eval[:, 1]
            data[:, 2] = pe.eval_function(mesh, u_full, x_eval, cell_eval)
        if rank == 0:
            os.makedirs(a.eval_out, exist_ok=True)
        for i in range(world):  # :597-605: one rank after the other appends to the same file
            if world > 1:
                dist.barrier()
            if rank == i:
                with open(os.path.join(a.eval_out, f"pressure_field_nproc{world}.txt"), "a") as f:
                    np.savetxt(f, data, fmt="%.8f", delimiter=",")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
