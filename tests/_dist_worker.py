"""One rank of a multi-process halo test (launched by tests/test_distributed.py).

argv: mode outdir P nx ny nz gx gy gz overlap
  mode "cpu": gloo backend, CPU tensors, oracle-backed kernels injected
  mode "gpu": gloo transport with host staging, HIP kernels on cuda:0 for every rank
              (a 1-GPU box cannot run RCCL between two ranks of one device)
Writes owned (lexicographic id, value) pairs of  y = K x  after fwd halo / apply / rev halo,
and of a forward-scattered vector, to outdir/rank<r>.npz.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import fusgpu_loader  # noqa: E402
from conftest import ref_field  # noqa: E402
from halo_cpu import OracleHaloKernels, global_cell_constants  # noqa: E402


class StagedComm:
    """gloo transport for device tensors (staged through the host) -- test only."""

    def __init__(self, inner):
        self.inner = inner
        self.rank, self.size, self.backend = inner.rank, inner.size, inner.backend

    def alltoallv(self, send, send_counts, recv, recv_counts, async_op=False):
        torch.cuda.synchronize()
        s, r = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        self.inner.alltoallv(s, send_counts, r, recv_counts)
        recv.copy_(r)
        return None

    def alltoallv_int64(self, *a):
        return self.inner.alltoallv_int64(*a)


def run_solver(outdir, P, cells, grid, overlap, nonlinear=False):
    """mode "gpu-solver" / "gpu-solver-nl": the fused linear / Westervelt RK4 solver on a
    partitioned mesh, all ranks on cuda:0."""
    rank = dist.get_rank()
    boxmesh, ls, scat = (fusgpu_loader.submodule(m) for m in ("boxmesh", "linear_solver", "scatterer"))
    nls = fusgpu_loader.submodule("nonlinear_solver")
    torch.cuda.set_device(0)
    L = 0.012
    mesh = boxmesh.BoxMesh(P, cells, grid=grid, rank=rank, length=L)
    serial = boxmesh.BoxMesh(P, cells, length=L)
    h = ls.time_step_parameters(serial, P, 1500.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    if nonlinear:
        solver = nls.WesterveltSpectral3D(mesh, np.float64, speed_of_sound=1500.0, source_frequency=0.5e6,
                                          comm=StagedComm(scat.TorchComm()), fused=True, overlap=bool(overlap))
    else:
        solver = ls.LinearSpectral3D(mesh, np.float64, comm=StagedComm(scat.TorchComm()), fused=True, overlap=bool(overlap))
    solver.init()
    solver.rk4(0.0, tf, dt, max_steps=8)
    torch.cuda.synchronize()
    lex = mesh.global_lexicographic_ids()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), lex_owned=lex[: mesh.nlocal], u_owned=solver.u_sol(), dt=dt)
    dist.barrier()
    dist.destroy_process_group()


def main():
    mode, outdir = sys.argv[1], sys.argv[2]
    P, nx, ny, nz, gx, gy, gz, overlap = (int(v) for v in sys.argv[3:11])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == gx * gy * gz
    if mode in ("gpu-solver", "gpu-solver-nl"):
        return run_solver(outdir, P, (nx, ny, nz), (gx, gy, gz), overlap, nonlinear=mode.endswith("nl"))
    boxmesh, gll, pre = (fusgpu_loader.submodule(m) for m in ("boxmesh", "gll", "precompute"))
    scat = fusgpu_loader.submodule("scatterer")
    mesh = boxmesh.BoxMesh(P, (nx, ny, nz), grid=(gx, gy, gz), rank=rank, perturb=0.16, seed=3)
    pts, wts, D = gll.tabulate_1d(P)
    n = P + 1
    w3 = gll.tensor_weights_3d(wts)
    dg = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))
    G = np.zeros((mesh.ncells, n**3, 6))
    pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), mesh.ncells, dg, w3)
    cc = global_cell_constants(mesh)
    x = ref_field(mesh.dof_coordinates())
    x[mesh.nlocal:] = -777.0  # ghosts are stale until the forward scatter
    comm = scat.TorchComm()
    if mode == "cpu":
        from oracle.oracle_c import OracleLib

        O = OracleLib()

        def apply_fn(x_, c_, y_, G_, d_):
            O.stiffness_apply(P, D, x_.numpy(), np.ascontiguousarray(c_.numpy()), y_.numpy(),
                              np.ascontiguousarray(G_.numpy()), np.ascontiguousarray(d_.numpy()))

        dev = torch.device("cpu")
        halo = scat.HaloApply(mesh, None, comm, np.float64, overlap=bool(overlap), kernels=OracleHaloKernels(), apply_fn=apply_fn)
    else:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        ops = fusgpu_loader.submodule("operators")
        op = ops.stiffness_operator(P, D.flatten(), np.float64)
        halo = scat.HaloApply(mesh, op, StagedComm(comm), np.float64, overlap=bool(overlap))
    x_d = torch.from_numpy(x).to(dev)
    y_d = torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev)
    halo.apply(x_d, torch.from_numpy(cc).to(dev), y_d, torch.from_numpy(G).to(dev), torch.from_numpy(mesh.dofmap).to(dev))
    if dev.type == "cuda":
        torch.cuda.synchronize()
    lex = mesh.global_lexicographic_ids()
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), lex_owned=lex[: mesh.nlocal], y_owned=y_d.cpu().numpy()[: mesh.nlocal],
             lex_all=lex, x_after_fwd=x_d.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
