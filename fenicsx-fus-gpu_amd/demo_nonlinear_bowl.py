#!/usr/bin/env python3
"""
Westervelt (nonlinear, attenuating) wave from a source face into a curved "bowl" geometry, explicit RK4, on the MI355X
operators -- the counterpart of the reference's cuda/demo_nonlinear_bowl.py: same physical parameters (:56-75), time-step
rule (CFL = 0.40, :119-125; final time L/c + 8/f), material coefficients (:357-374), stage sequence (:540-650: lumped
mass with the nonlinear term, two stiffness applies, mass of v_n^2, g and dg/dt source terms, absorbing facets) and the
collection of the pressure field over the LAST PERIOD (:662-680: once t > L/c + 6/f, one dump per time step for one
period), on a synthetic mesh: the reference reads the H131 transducer mesh from XDMF (absent from its repository); here
a structured box is warped by a smooth bowl map, which gives what its geometry gives the kernels -- trilinear, NON-affine
cells (G varies per quadrature point; P1 geometry, cuda/demo_nonlinear_bowl.py:317).

    python fenicsx-fus-gpu_amd/demo_nonlinear_bowl.py [--degree 6] [--cells N] [--out-dir DIR] [--max-steps K]
    python -m torch.distributed.run --nproc-per-node 8 fenicsx-fus-gpu_amd/demo_nonlinear_bowl.py

Dumps: ``DIR/pressure_field_<k>.txt`` for k = 0 .. steps_per_period-1, rows ``x,y,p`` on the mid-z plane of the dof grid
(the reference evaluates ``u_n.eval`` at its own point set and writes x, y, value rows with the same format string);
with several ranks every rank appends its owned points, as the reference's ranks append theirs.
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
    ap.add_argument("--degree", type=int, default=6)
    ap.add_argument("--cells", type=int, default=None, help="cells per direction of the whole box (default: 2 per wavelength at P = 6 scale)")
    ap.add_argument("--length", type=float, default=None, help="domain length in m (reference: 0.08; default here 0.012 so the default run takes seconds)")
    ap.add_argument("--reference-sequence", action="store_true", help="the reference's unfused launch sequence (four cell kernels per stage)")
    ap.add_argument("--geometry", default="auto", choices=["auto", "kernel", "array"],
                    help="auto: G formed in the cell kernel from the vertices for the fused stage of degree >= 3 (the solver's default); "
                         "array: the reference's precomputed G array; kernel: force the kernel form")
    ap.add_argument("--in-kernel-geometry", action="store_true", help="same as --geometry kernel")
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--out-dir", default=None, help="write the last-period pressure fields there")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    import fusgpu_loader

    boxmesh, ls, nls, scat = (fusgpu_loader.submodule(m) for m in ("boxmesh", "linear_solver", "nonlinear_solver", "scatterer"))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    comm = None
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        comm = scat.default_comm()

    # cuda/demo_nonlinear_bowl.py:56-75
    float_type = np.float64
    speed_of_sound, density = 1480.0, 1000.0
    source_frequency = 1.1e6
    source_velocity = 0.38557513826589934
    source_amplitude = density * speed_of_sound * source_velocity
    period = 1.0 / source_frequency
    nonlinear_coefficient, attenuation_coefficient_dB = 3.5, 0.2
    domain_length = a.length if a.length is not None else 0.012
    wave_length = speed_of_sound / source_frequency
    P = a.degree
    num_element = a.cells if a.cells is not None else max(2, int(2 * domain_length / wave_length))
    grid = boxmesh.default_grid(world)
    L = domain_length

    def bowl(xg):  # smooth map of the box: the source face x = 0 becomes a shallow bowl, the far face stays plane
        out = xg.copy()
        yy, zz = xg[:, 1] / L - 0.5, xg[:, 2] / L - 0.5
        out[:, 0] = xg[:, 0] + 0.15 * (L / num_element) * 4 * (yy * yy + zz * zz) * (1.0 - xg[:, 0] / L)
        return out

    mesh = boxmesh.BoxMesh(P, num_element, grid=grid, rank=rank, length=L, dtype=float_type, warp=bowl)
    h = ls.time_step_parameters(mesh, P, speed_of_sound, source_frequency, L)
    if world > 1:
        hm = torch.tensor([h], dtype=torch.float64, device="cuda")
        dist.all_reduce(hm, op=dist.ReduceOp.MIN)  # comm.Allreduce(hmin, mesh_size, op=MPI.MIN), :108
        h = float(hm.item())
    # :119-125 (CFL 0.40, an integer number of steps per period, final time L/c + 8/f)
    CFL = 0.40
    dt = CFL * h / (speed_of_sound * P**2)
    step_per_period = int(period / dt) + 1
    dt = period / step_per_period
    tf = L / speed_of_sound + 8.0 / source_frequency
    nstep = int(tf / dt) + 1
    if rank == 0:
        print(f"Number of steps: {nstep}", flush=True)
        print(f"Number of steps per period: {step_per_period}", flush=True)
        print(f"Number of degrees-of-freedom: {mesh.ndofs_global}", flush=True)
    solver = nls.WesterveltSpectral3D(mesh, float_type, speed_of_sound, density, source_frequency, source_amplitude,
                                      nonlinear_coefficient, attenuation_coefficient_dB, comm=comm, fused=not a.reference_sequence,
                                      in_kernel_geometry=True if (a.in_kernel_geometry or a.geometry == "kernel") else ("auto" if a.geometry == "auto" else False))
    solver.init()

    # sampling set: the owned dofs on the mid-z plane of the global dof grid
    lex = mesh.global_lexicographic_ids()[: mesh.nlocal]
    gd = mesh.global_dof_dims
    on_plane = np.nonzero((lex % gd[2]) == gd[2] // 2)[0]
    xyz = mesh.dof_coordinates()[: mesh.nlocal][on_plane]
    data = np.zeros((on_plane.size, 3))
    data[:, 0], data[:, 1] = xyz[:, 0], xyz[:, 1]
    if a.out_dir and rank == 0:
        os.makedirs(a.out_dir, exist_ok=True)
    if world > 1:
        dist.barrier()

    if rank == 0:
        print("Solve!", flush=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_collect = L / speed_of_sound + 6.0 / source_frequency  # :662
    budget = a.max_steps if a.max_steps is not None else nstep
    # up to the collection window in one go (no host round trip per step), then step by step with one dump per step
    n_before = min(budget, max(0, int(np.floor(t_collect / dt)) - 1))  # t stays <= the threshold: no dump is skipped
    t, steps = solver.rk4(0.0, tf, dt, max_steps=n_before) if n_before > 0 else (0.0, 0)
    step_period = 0
    while t < tf and steps < budget:
        t, more = solver.rk4(t, tf, dt, max_steps=1)
        steps += more
        if steps % 100 == 0 and rank == 0:
            print(f"t: {t:5.5},\t Steps: {steps}/{nstep}", flush=True)
        if t > t_collect and step_period < step_per_period:
            if a.out_dir:
                data[:, 2] = solver.u_sol()[on_plane]
                with open(os.path.join(a.out_dir, f"pressure_field_{step_period}.txt"), "a") as f:
                    np.savetxt(f, data, fmt="%.8f", delimiter=",")  # the reference's format, :675
            step_period += 1
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if rank == 0:
        print(f"t: {t:5.5},\t Steps: {steps}/{nstep}", flush=True)
        print(f"Fields collected over the last period: {step_period}/{step_per_period}")
        print(f"Solve time: {el}")
        print(f"Solve time per step: {el / max(steps, 1)}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
