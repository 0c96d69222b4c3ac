"""One rank of an ``mpirun``-style world: the communicator handed to the package is what the reference's drivers hand to theirs --
an ``MPI.Comm`` (cuda/demo_linear_box.py:41 ``comm = MPI.COMM_WORLD``; cuda/scatterer.py:104-110, 191-197 ``comm: MPI.Comm``) -- here
the file-backed stand-in of tests/fake_mpi.py (mpi4py is not in this image), across REAL processes sharing cuda:0.  No
torch.distributed process group exists in these processes: the halo data moves through libfusgpu.so's PEER transport, the MPI
communicator carries the bootstrap (arena handles, votes, the index exchange of compute_scatterer_data) only.

    python tests/_mpi_worker.py <mode> <rank> <world> <rendezvous dir> <args...>

modes
  golden <fixture.npz> <dtype>   the reference's call sequence verbatim: ``compute_scatterer_data(index_map)`` with the world
                                 communicator as default, ``scatter_forward / scatter_reverse(comm, owners_data, ghosts_data, N,
                                 float_type)`` with the RAW communicator; closures' inputs -> the reference closures' outputs
  solver <rk4 fixture.npz> <fused>  ``LinearSpectral3D(mesh, comm=MPI communicator)`` == the reference-driven loop's fixture
  badcomm                        anything that is neither a package communicator nor an MPI one raises TypeError
Prints MPI_WORKER_OK <rank> on success; any failure is a non-zero exit."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode, rank, world, rdv = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    args = sys.argv[5:]
    os.environ.setdefault("FUS_IPC_SPIN_SECONDS", "10")
    import torch
    import torch.distributed as dist

    from conftest import pkg, rel_l2
    from fake_mpi import FileComm

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    comm = FileComm(rdv, rank, world)  # the stand-in for MPI.COMM_WORLD
    scat, boxmesh, utils, boot = pkg("scatterer"), pkg("boxmesh"), pkg("utils"), pkg("mpi_bootstrap")
    assert boot.is_mpi_comm(comm)
    # compute_scatterer_data(index_map) takes no communicator in the reference (cuda/utils.py:8: MPI.COMM_WORLD is hard-coded);
    # without mpi4py the stand-in is registered as the world communicator the same way mpi4py's would be found
    boot.world_if_available = lambda: comm

    if mode == "golden":
        d = np.load(args[0])
        dtype = np.dtype(args[1]).type
        P, shape, grid = int(d["P"]), tuple(int(v) for v in d["shape"]), tuple(int(v) for v in d["grid"])
        assert int(np.prod(grid)) == world
        m = boxmesh.BoxMesh(P, shape, grid=grid, rank=rank)
        owners_data, ghosts_data = utils.compute_scatterer_data(m.index_map)  # cuda/demo_linear_box.py:192
        assert comm.calls["alltoall"] == 1  # the index exchange went over the MPI communicator
        # the same plan as the all-ranks-in-one-process builder the other tests use
        meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r) for r in range(world)]
        od, gd = utils.compute_scatterer_data_all([mm.index_map for mm in meshes])
        for got, ref in ((utils.to_flat(owners_data), od[rank]), (utils.to_flat(ghosts_data), gd[rank])):
            assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(got, ref))
        scatter_fwd = scat.scatter_forward(comm, owners_data, ghosts_data, m.nlocal, dtype)  # cuda/demo_linear_box.py:206-207
        scatter_rev = scat.scatter_reverse(comm, owners_data, ghosts_data, m.nlocal, dtype)
        assert type(scatter_fwd).__name__ == "_NativeScatter" and scatter_fwd.comm.transport == "peer" and scatter_fwd.comm.size == world
        assert scatter_fwd.comm is scatter_rev.comm  # one library communicator per MPI communicator
        for kind, sc in (("fwd", scatter_fwd), ("rev", scatter_rev)):
            buf = torch.from_numpy(d[f"in_{rank}"].astype(dtype)).to(dev)
            for _ in range(2 if kind == "fwd" else 1):
                sc(buf)
            torch.cuda.synchronize()
            ref, got = d[f"ref_{kind}_{rank}"], buf.cpu().numpy()
            if kind == "fwd":
                assert np.array_equal(got, ref.astype(dtype)), f"forward, rank {rank}"
            else:
                assert np.allclose(got, ref, rtol=0, atol=1e-13 if dtype == np.float64 else 2e-6), f"reverse, rank {rank}"
        v = torch.from_numpy(d[f"in_{rank}"].astype(dtype)).to(dev)
        for _ in range(20):
            scatter_fwd(v)
        torch.cuda.synchronize()
        assert np.array_equal(v.cpu().numpy(), d[f"ref_fwd_{rank}"].astype(dtype))
        assert scatter_fwd.status()["timeouts"] == 0 and scatter_rev.status()["timeouts"] == 0
        comm.barrier()  # nobody frees an arena a neighbour may still write a credit into
        scatter_fwd.close(), scatter_rev.close()
    elif mode == "solver":
        d = np.load(args[0])
        fused = bool(int(args[1]))
        ls = pkg("linear_solver")
        P, shape, grid = int(d["P"]), tuple(int(v) for v in d["shape"]), tuple(int(v) for v in d["grid"])
        assert int(np.prod(grid)) == world
        mesh = boxmesh.BoxMesh(P, shape, grid=grid, rank=rank, length=tuple(float(v) for v in d["lengths"]), perturb=float(d["perturb"]),
                               seed=int(d["seed"]))
        s = ls.LinearSpectral3D(mesh, np.float64, speed_of_sound=float(d["c0"]), density=float(d["rho0"]), source_frequency=float(d["f0"]),
                                source_amplitude=float(d["p0"]), comm=comm, fused=fused)
        s.init()
        nsteps, dt = int(d["nsteps"]), float(d["dt"])
        _, steps = s.rk4(0.0, 1.0, dt, max_steps=nsteps)
        torch.cuda.synchronize()
        assert steps == nsteps and s.halo.schedule_kind == "concurrent" and s.halo.health() == 0
        eu = rel_l2(s.u_sol(), d[f"ref_u_tn_{rank}"][: mesh.nlocal])
        ev = rel_l2(s.v_sol(), d[f"ref_v_tn_{rank}"][: mesh.nlocal])
        assert eu < 1e-11 and ev < 1e-11, f"rank {rank}: u {eu} v {ev} vs the reference-driven loop"
        assert rel_l2(s.u_sol(with_ghosts=True), d[f"ref_u_tn_{rank}"]) < 1e-11
        comm.barrier()
        del s
    elif mode == "badcomm":
        m = boxmesh.BoxMesh(2, (4, 2, 2), grid=(2, 1, 1), rank=rank)
        meshes = [boxmesh.BoxMesh(2, (4, 2, 2), grid=(2, 1, 1), rank=r) for r in range(2)]
        od, gd = utils.compute_scatterer_data_all([mm.index_map for mm in meshes])
        for bad in (object(), "MPI.COMM_WORLD", 0):
            try:
                scat.scatter_forward(bad, od[rank], gd[rank], m.nlocal, np.float64)
            except TypeError as e:
                assert "NativeComm" in str(e) and "MPI" in str(e), str(e)
            else:
                raise AssertionError(f"scatter_forward accepted {bad!r}")
    else:
        raise SystemExit(f"unknown mode {mode}")
    assert not dist.is_initialized(), "an mpirun world needs no torch.distributed process group"
    print(f"MPI_WORKER_OK {rank}", flush=True)


if __name__ == "__main__":
    main()
