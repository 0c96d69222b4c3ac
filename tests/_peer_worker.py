"""One rank of a multi-PROCESS world that shares cuda:0 (the pool has one GPU per box; on an 8-GPU node each rank has its
own).  Halo transport = PEER (csrc/halo_ipc.hpp): arenas mapped across processes with HIP IPC handles, exchanges made of
the send / receive kernels only.  torch.distributed (gloo) carries nothing but the one-off bootstrap.

    python tests/_peer_worker.py <mode> <rank> <world> <port> <args...>

modes
  golden <fixture.npz> <dtype>     forward (twice) and reverse scatter of the reference closures' inputs == their outputs
  apply <P> <nx> <ny> <nz> <gx> <gy> <gz> <ghost_order> <schedule>
                                   HaloApply (two applies) on a partitioned perturbed box == the serial C oracle
  solver <rk4 fixture.npz> <fused 0|1>
                                   the linear RK4 solver, one PROCESS per rank, == what the reference's own operators and
                                   scatter closures produced in the same loop (tests/golden/rk4_*_2ranks.npz)
  soak <iterations>                forward + reverse exchanges with data that changes every iteration and random host-side
                                   skew between the ranks: every ghost value and every owned sum checked exactly each time
  deadpeer                         the owner rank never posts its exchange: the ghosting rank's receive must give up after
                                   FUS_IPC_SPIN_SECONDS and report a time-out instead of hanging
  solver_deadpeer                  the linear solver on two processes; rank 1 stops stepping after 2 steps: rank 0's rk4() must
                                   RAISE (not return a field), and rank 1, stepping again later, must raise too (poisoned flags)
  hybrid <P> <nx> <ny> <nz> <gx> <gy> <gz> <ghost_order> <ranks_per_process> <apply|solver>
                                   a world of gx*gy*gz ranks on world = that / ranks_per_process PROCESSES, each driving its
                                   ranks in lock step (the 8-rank 2x2x2 partition on 4 processes: rank 0 ghosted by 7 ranks,
                                   rank 7 ghosting from 7; face / edge / corner segments; neighbours of the same process
                                   through plain pointers, the others through HIP IPC mappings).  apply: HaloApply == the
                                   serial C oracle; solver: the fused linear solver on the partition == the one-rank solver
Prints PEER_WORKER_OK <rank> on success; any failure is a non-zero exit."""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode, rank, world, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    args = sys.argv[5:]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault("FUS_IPC_SPIN_SECONDS", {"deadpeer": "1", "solver_deadpeer": "2", "hybrid": "60"}.get(mode, "10"))
    # (hybrid: 4 processes x 2 ranks x (launch + exchange stream) on ONE card is more hardware queues than the card has, so the
    # processes' queues are time-sliced and a hand-off between two processes can cost scheduling quanta, not microseconds)
    import torch
    import torch.distributed as dist

    import fusgpu_loader
    from conftest import build_problem, pkg, ref_field, rel_l2
    from halo_cpu import global_cell_constants

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scat, boxmesh, utils = pkg("scatterer"), pkg("boxmesh"), pkg("utils")
    if mode == "hybrid":
        hybrid(args, rank, world, dev)
        dist.barrier()
        dist.destroy_process_group()
        print(f"PEER_WORKER_OK {rank}", flush=True)
        return
    comm = scat.NativeComm(transport="peer")
    assert comm.backend == "peer" and comm.size == world and comm.rank == rank

    if mode == "golden":
        d = np.load(args[0])
        dtype = np.dtype(args[1]).type
        P, shape, grid = int(d["P"]), tuple(int(v) for v in d["shape"]), tuple(int(v) for v in d["grid"])
        assert int(np.prod(grid)) == world
        meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r) for r in range(world)]
        od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
        m = meshes[rank]
        fwd = scat.scatter_forward(comm, od[rank], gd[rank], m.nlocal, dtype)
        rev = scat.scatter_reverse(comm, od[rank], gd[rank], m.nlocal, dtype)
        for kind, sc in (("fwd", fwd), ("rev", rev)):
            buf = torch.from_numpy(d[f"in_{rank}"].astype(dtype)).to(dev)
            for _ in range(2 if kind == "fwd" else 1):  # forward is idempotent: the second one reuses arena + credits
                sc(buf)
            torch.cuda.synchronize()
            ref, got = d[f"ref_{kind}_{rank}"], buf.cpu().numpy()
            if kind == "fwd":
                assert np.array_equal(got, ref.astype(dtype)), f"forward, rank {rank}"
            else:
                assert np.allclose(got, ref, rtol=0, atol=1e-13 if dtype == np.float64 else 2e-6), f"reverse, rank {rank}"
        # many exchanges back to back: credits and sequence flags under load
        v = torch.from_numpy(d[f"in_{rank}"].astype(dtype)).to(dev)
        for _ in range(50):
            fwd(v)
        torch.cuda.synchronize()
        assert np.array_equal(v.cpu().numpy(), d[f"ref_fwd_{rank}"].astype(dtype))
        assert fwd.status()["timeouts"] == 0 and rev.status()["timeouts"] == 0
        dist.barrier()  # nobody frees an arena a neighbour may still write a credit into
        fwd.close(), rev.close()
    elif mode == "apply":
        P = int(args[0])
        cells, grid = tuple(int(v) for v in args[1:4]), tuple(int(v) for v in args[4:7])
        ghost_order = args[7] if args[7] in ("owner", "lex") else int(args[7])
        schedule = args[8]
        assert int(np.prod(grid)) == world
        from oracle.oracle_c import OracleLib

        ops, gll, pre = pkg("operators"), pkg("gll"), pkg("precompute")
        pts, wts, D = gll.tabulate_1d(P)
        n = P + 1
        mesh = boxmesh.BoxMesh(P, cells, grid=grid, rank=rank, perturb=0.16, seed=3, ghost_order=ghost_order)
        G = np.zeros((mesh.ncells, n**3, 6))
        pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), mesh.ncells,
                                              pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts)), gll.tensor_weights_3d(wts))
        x = ref_field(mesh.dof_coordinates())
        x[mesh.nlocal:] = -777.0  # ghosts are stale until the forward scatter
        x_d, y_d = torch.from_numpy(x).to(dev), torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev)
        cc_d, G_d = torch.from_numpy(global_cell_constants(mesh)).to(dev), torch.from_numpy(G).to(dev)
        dm_d = torch.from_numpy(mesh.dofmap).to(dev)
        op = ops.stiffness_operator(P, D.flatten(), np.float64)
        halo = scat.HaloApply(mesh, op, comm, np.float64, schedule=schedule)  # plan: index exchange over gloo
        assert halo.schedule_kind == schedule
        for _ in range(2):
            y_d.zero_()
            halo.apply(x_d, cc_d, y_d, G_d, dm_d)
        torch.cuda.synchronize()
        assert halo.health() == 0
        pb = build_problem(P, cells, perturb=0.16, seed=3)
        ms = pb["mesh"]
        y_ser = np.zeros(ms.ndofs)
        OracleLib().stiffness_apply(P, pb["D"], pb["x"], global_cell_constants(ms), y_ser, pb["G"], ms.dofmap)
        lex = mesh.global_lexicographic_ids()
        err = rel_l2(y_d.cpu().numpy()[: mesh.nlocal], y_ser[lex[: mesh.nlocal]])
        assert err < 1e-12, f"rank {rank}: partitioned apply vs serial oracle {err}"
        assert np.allclose(x_d.cpu().numpy(), pb["x"][lex], rtol=0, atol=1e-12), "ghosts not refreshed"
        dist.barrier()
        halo.fwd.close(), halo.rev.close()
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
        full = s.u_sol(with_ghosts=True)  # forward scatter at the end, as the demo does before copying to the host
        assert rel_l2(full, d[f"ref_u_tn_{rank}"]) < 1e-11
        dist.barrier()
        del s
    elif mode == "soak":
        import time

        iters = int(args[0])
        P, shape, grid = 2, (4, 4, 2), (2, 2, 1)
        assert world == 4
        meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r, ghost_order=5) for r in range(world)]
        od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
        m = meshes[rank]
        fwd = scat.scatter_forward(comm, od[rank], gd[rank], m.nlocal, np.float64)
        rev = scat.scatter_reverse(comm, od[rank], gd[rank], m.nlocal, np.float64)
        lex = torch.from_numpy(m.global_lexicographic_ids().astype(np.float64)).to(dev)
        # how many ranks ghost each of my owned dofs (reverse: every ghosting rank adds k to the owner's entry)
        mult = np.zeros(m.ndofs)
        np.add.at(mult, np.asarray(gd[rank][0], dtype=np.int64), 1.0)
        mult_d = torch.from_numpy(mult).to(dev)
        rng = np.random.default_rng(rank)
        bad = torch.zeros((), dtype=torch.float64, device=dev)
        for k in range(1, iters + 1):
            if rng.random() < 0.05:
                time.sleep(float(rng.random()) * 0.004)  # host-side skew: this rank posts late now and then
            v = lex * 3.0 + float(k)  # integer-valued: every comparison below is exact
            v[m.nlocal:] = -1.0
            fwd(v)
            bad += (v != lex * 3.0 + float(k)).sum()
            w = torch.zeros_like(lex)
            w[m.nlocal:] = float(k)
            rev(w)
            bad += (w[: m.nlocal] != mult_d[: m.nlocal] * float(k)).sum()
        torch.cuda.synchronize()
        assert float(bad.item()) == 0.0, f"rank {rank}: {int(bad.item())} wrong values in {iters} iterations"
        assert fwd.status()["timeouts"] == 0 and rev.status()["timeouts"] == 0
        assert fwd.status()["forward_posted"] == iters and rev.status()["reverse_posted"] == iters
        dist.barrier()
        fwd.close(), rev.close()
    elif mode == "deadpeer":
        import time

        m = boxmesh.BoxMesh(2, (4, 2, 2), grid=(2, 1, 1), rank=rank)
        meshes = [boxmesh.BoxMesh(2, (4, 2, 2), grid=(2, 1, 1), rank=r) for r in range(2)]
        od, gd = utils.compute_scatterer_data_all([mm.index_map for mm in meshes])
        fwd = scat.scatter_forward(comm, od[rank], gd[rank], m.nlocal, np.float64)  # connects (collective)
        buf = torch.zeros(m.ndofs, dtype=torch.float64, device=dev)
        if rank == 1:  # the ghosting rank posts; its owner (rank 0) stays silent
            t0 = time.perf_counter()
            fwd(buf)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            st = fwd.status()
            assert st["timeouts"] >= 1, st
            assert 0.5 < el < 8.0, el  # FUS_IPC_SPIN_SECONDS = 1 for this mode: gave up, did not hang
            t0 = time.perf_counter()
            fwd(buf)  # after a time-out the halo no longer waits at all
            torch.cuda.synchronize()
            assert time.perf_counter() - t0 < 0.5
        dist.barrier()
        fwd.close()
    elif mode == "solver_deadpeer":
        import time

        lib_mod = pkg("_lib")
        d = np.load(os.path.join(ROOT, "tests", "golden", "rk4_P2_4x2x2_pert_2ranks.npz"))
        ls = pkg("linear_solver")
        P, shape, grid = int(d["P"]), tuple(int(v) for v in d["shape"]), tuple(int(v) for v in d["grid"])
        mesh = boxmesh.BoxMesh(P, shape, grid=grid, rank=rank, length=tuple(float(v) for v in d["lengths"]), perturb=float(d["perturb"]),
                               seed=int(d["seed"]))
        s = ls.LinearSpectral3D(mesh, np.float64, speed_of_sound=float(d["c0"]), density=float(d["rho0"]), source_frequency=float(d["f0"]),
                                source_amplitude=float(d["p0"]), comm=comm, fused=True)
        s.init()
        dt = float(d["dt"])
        if rank == 0:
            t0 = time.perf_counter()
            try:
                s.rk4(0.0, 1.0, dt, max_steps=6)  # the neighbour leaves the loop after 2 steps
            except lib_mod.FusGpuError as e:
                assert "INVALID" in str(e) and "rank 0" in str(e), str(e)
            else:
                raise AssertionError("rk4() returned a field although the neighbour stopped exchanging")
            assert time.perf_counter() - t0 < 20.0  # one bounded wait (2 s), then no waiting at all
            st = s.halo.rev.status()
            assert s.halo.health() >= 1 and (st["timeouts"] >= 1 or s.halo.fwd.status()["timeouts"] >= 1 or s.fwd_v.status()["timeouts"] >= 1)
            dist.barrier()  # (A) rank 0 has failed
            dist.barrier()  # (B) rank 1 has seen it
        else:
            _, steps = s.rk4(0.0, 1.0, dt, max_steps=2)  # healthy so far: every exchange had its partner
            assert steps == 2 and s.halo.health() == 0
            dist.barrier()  # (A)
            try:
                s.rk4(2 * dt, 1.0, dt, max_steps=1)  # the first flag it reads from rank 0 is poisoned
            except lib_mod.FusGpuError as e:
                assert "INVALID" in str(e), str(e)
            else:
                raise AssertionError("the neighbour's failure did not reach this rank")
            st = [sc.status() for sc in (s.halo.fwd, s.halo.rev, s.fwd_v)]
            assert sum(x["poisoned"] for x in st) >= 1 and sum(x["timeouts"] for x in st) == 0, st
            dist.barrier()  # (B)
        del s
    else:
        raise SystemExit(f"unknown mode {mode}")
    comm.close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"PEER_WORKER_OK {rank}", flush=True)


def hybrid(args, proc, nproc, dev):
    """``ranks_per_process`` ranks of the world in each process, driven in lock step (generators), PEER transport."""
    import torch
    import torch.distributed as dist

    from conftest import build_problem, pkg, ref_field, rel_l2
    from halo_cpu import global_cell_constants
    from oracle.oracle_c import OracleLib

    scat, boxmesh, utils, ops, gll, pre, ls = (pkg(m) for m in ("scatterer", "boxmesh", "utils", "operators", "gll", "precompute", "linear_solver"))
    P = int(args[0])
    cells, grid = tuple(int(v) for v in args[1:4]), tuple(int(v) for v in args[4:7])
    ghost_order = args[7] if args[7] in ("owner", "lex") else int(args[7])
    rpp = int(args[8])
    R = int(np.prod(grid))
    assert R == nproc * rpp
    mine = list(range(proc * rpp, (proc + 1) * rpp))
    meshes = [boxmesh.BoxMesh(P, cells, grid=grid, rank=r, perturb=0.16, seed=3, ghost_order=ghost_order) for r in range(R)]
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    # what the 2x2x2 partition is about: rank 0 is ghosted by 7 ranks, rank 7 ghosts dofs of 7 owners (3 faces, 3 edges, 1
    # corner: segments of many chunks next to one-element ones)
    if grid == (2, 2, 2):
        assert len(gd[0][3]) == 7 and len(od[R - 1][3]) == 7 and min(int(v) for v in od[R - 1][1]) == 1
    what = args[9]

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

    if what == "apply":
        pts, wts, D = gll.tabulate_1d(P)
        n = P + 1
        w3, dg = gll.tensor_weights_3d(wts), pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))
        op = ops.stiffness_operator(P, D.flatten(), np.float64)
        comms, halos, ranks = [], [], []
        for r in mine:
            mesh = meshes[r]
            G = np.zeros((mesh.ncells, n**3, 6))
            pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), mesh.ncells, dg, w3)
            x = ref_field(mesh.dof_coordinates())
            x[mesh.nlocal:] = -777.0
            ranks.append(dict(mesh=mesh, x=torch.from_numpy(x).to(dev), y=torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev),
                              cc=torch.from_numpy(global_cell_constants(mesh)).to(dev), G=torch.from_numpy(G).to(dev),
                              dm=torch.from_numpy(mesh.dofmap).to(dev)))
            comms.append(scat.NativeComm(local=(4242, R, r), transport="peer", hosted=mine))
        for r, rk, comm in zip(mine, ranks, comms):  # closures in the same order in every process: rank-local index 0, 1, ...
            halos.append(scat.HaloApply(rk["mesh"], op, comm, np.float64, plan=(od[r], gd[r]), schedule="concurrent"))

        for rep in range(3):
            for rk in ranks:
                rk["y"].zero_()
            lockstep([h.apply_schedule(rk["x"], rk["cc"], rk["y"], rk["G"], rk["dm"]) for h, rk in zip(halos, ranks)])
        torch.cuda.synchronize()
        assert all(h.health() == 0 for h in halos)
        pb = build_problem(P, cells, perturb=0.16, seed=3)
        ms = pb["mesh"]
        y_ser = np.zeros(ms.ndofs)
        OracleLib().stiffness_apply(P, pb["D"], pb["x"], global_cell_constants(ms), y_ser, pb["G"], ms.dofmap)
        for r, rk in zip(mine, ranks):
            m = rk["mesh"]
            lex = m.global_lexicographic_ids()
            err = rel_l2(rk["y"].cpu().numpy()[: m.nlocal], y_ser[lex[: m.nlocal]])
            assert err < 1e-12, f"rank {r}: partitioned apply vs serial oracle {err}"
            assert np.allclose(rk["x"].cpu().numpy(), pb["x"][lex], rtol=0, atol=1e-12), f"rank {r}: ghosts not refreshed"
        dist.barrier()
        del halos
        return
    if what == "westervelt":
        # the fused Westervelt solver (BASELINE config 5's loop: set-up scatter of the three assembled diagonals, per stage the
        # grouped forward scatter of (w, v_n) and the reverse of b) on the partition == the one-rank solver
        nls = pkg("nonlinear_solver")
        L = 0.012
        geom = len(args) > 10 and args[10] == "geom"
        kw = dict(speed_of_sound=1500.0, source_frequency=0.5e6, fused=True, in_kernel_geometry=geom)
        wmesh = [boxmesh.BoxMesh(P, cells, grid=grid, rank=r, length=L, perturb=0.1, seed=6, ghost_order=ghost_order) for r in range(R)]
        odw, gdw = utils.compute_scatterer_data_all([m.index_map for m in wmesh])
        commsw = [scat.NativeComm(local=(4444, R, r), transport="peer", hosted=mine) for r in mine]
        solvers = [nls.WesterveltSpectral3D(wmesh[r], np.float64, comm=c, halo_plan=(odw[r], gdw[r]), defer_setup_exchange=True, **kw)
                   for r, c in zip(mine, commsw)]
        lockstep([s_._setup for s_ in solvers])
        one = boxmesh.BoxMesh(P, cells, length=L, perturb=0.1, seed=6)
        h = ls.time_step_parameters(one, P, 1500.0, 0.5e6, L)
        dts, tf, _ = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
        for s_ in solvers:
            s_.init()
        lockstep([s_.rk4_schedule(0.0, tf, dts, max_steps=4) for s_ in solvers])
        torch.cuda.synchronize()
        for s_ in solvers:
            s_.check_halo_health("hybrid Westervelt solver")
        assert all(s_.halo.schedule_kind == "concurrent" for s_ in solvers)
        ref = nls.WesterveltSpectral3D(one, np.float64, **kw)
        ref.init()
        ref.rk4(0.0, tf, dts, max_steps=4)
        u_ref, v_ref = ref.u_sol(), ref.v_sol()
        assert np.abs(u_ref).max() > 0
        for r, s_ in zip(mine, solvers):
            lex = wmesh[r].global_lexicographic_ids()[: wmesh[r].nlocal]
            eu, ev = rel_l2(s_.u_sol(), u_ref[lex]), rel_l2(s_.v_sol(), v_ref[lex])
            assert eu < 1e-11 and ev < 1e-11, f"rank {r}: partitioned Westervelt solver vs one-rank solver u {eu} v {ev}"
        dist.barrier()
        del solvers
        return
    # the fused linear solver on the same partition (set-up reverse scatter, grouped forward scatters of (u_n, v_n), reverse
    # of b, concurrent schedule with the fork / join folded into the exchange kernels) == the one-rank solver
    L = (0.012, 0.012, 0.012)
    kw = dict(speed_of_sound=1500.0, density=1000.0, source_frequency=0.5e6, source_amplitude=60000.0)
    smesh = [boxmesh.BoxMesh(P, cells, grid=grid, rank=r, length=L, perturb=0.12, seed=5, ghost_order=ghost_order) for r in range(R)]
    od2, gd2 = utils.compute_scatterer_data_all([m.index_map for m in smesh])
    comms2 = [scat.NativeComm(local=(4343, R, r), transport="peer", hosted=mine) for r in mine]
    solvers = [ls.LinearSpectral3D(smesh[r], np.float64, comm=c, fused=True, halo_plan=(od2[r], gd2[r]), defer_setup_exchange=True, **kw)
               for r, c in zip(mine, comms2)]
    lockstep([s_._setup for s_ in solvers])
    one = boxmesh.BoxMesh(P, cells, length=L, perturb=0.12, seed=5)
    h = ls.time_step_parameters(one, P, 1500.0, 0.5e6, L[0])
    dts, tf, _ = ls.snap_time_step(h, P, 1500.0, 0.5e6, L[0])
    for s_ in solvers:
        s_.init()
    lockstep([s_.rk4_schedule(0.0, tf, dts, max_steps=5) for s_ in solvers])
    torch.cuda.synchronize()
    for s_ in solvers:
        s_.check_halo_health("hybrid solver")
    assert all(s_.halo.schedule_kind == "concurrent" for s_ in solvers)
    ref = ls.LinearSpectral3D(one, np.float64, fused=True, **kw)
    ref.init()
    ref.rk4(0.0, tf, dts, max_steps=5)
    u_ref = ref.u_sol()
    assert np.abs(u_ref).max() > 0
    for r, s_ in zip(mine, solvers):
        lex = smesh[r].global_lexicographic_ids()[: smesh[r].nlocal]
        e = rel_l2(s_.u_sol(), u_ref[lex])
        assert e < 1e-11, f"rank {r}: partitioned solver vs one-rank solver {e}"
    dist.barrier()
    del solvers


if __name__ == "__main__":
    main()
