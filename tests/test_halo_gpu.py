"""GPU parity of the halo path (SURVEY 8 rows a8 / a9 / b):

  * the four pack / unpack kernels (cuda/scatterer.py:18-101) against what the reference's own
    scatter closures produced on the same partitions (tests/golden/scatter_*.npz), f64 and f32;
  * the native exchange behind the C ABI (fus_halo_*, csrc/halo_comm.hpp) on the same fixtures,
    all ranks in this process (in-process transport: stream-ordered device copies, no host sync),
    owner-grouped ghosts (direct mode) and arbitrary ghost numberings (unpack_fwd / pack_rev);
  * the overlapped partitioned apply (HaloApply) with 2 / 4 / 8 ranks sharing cuda:0, ghosts NOT
    numbered owner by owner, asynchronous transport, against the serial oracle;
  * the RCCL transport itself in a 1-rank world: a rank that is its own neighbour
    (grouped ncclSend / ncclRecv to self).
RCCL refuses two ranks on one device, so N > 1 RCCL cannot run on a one-GPU box.

The PEER transport (csrc/halo_ipc.hpp: peer-mapped arenas, send / receive kernels, sequence flags) runs the same
in-process cases and, with 2 and 4 REAL processes sharing cuda:0 (arenas mapped through HIP IPC handles), the golden
scatter fixtures and the partitioned apply against the serial oracle."""

import itertools

import numpy as np
import pytest

from conftest import build_problem, golden_files, pkg, rel_l2, ref_field
from halo_cpu import global_cell_constants

pytestmark = pytest.mark.gpu

_world_ids = itertools.count(1000)


@pytest.fixture(scope="module")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    torch.cuda.set_device(0)
    return torch


def _partition(d, ghost_order="owner"):
    boxmesh, utils = pkg("boxmesh"), pkg("utils")
    P, shape, grid = int(d["P"]), tuple(int(v) for v in d["shape"]), tuple(int(v) for v in d["grid"])
    R = int(np.prod(grid))
    meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r, ghost_order=ghost_order) for r in range(R)]
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    return meshes, od, gd


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("path", golden_files("scatter_"), ids=lambda p: p.split("/")[-1][:-4])
def test_pack_unpack_kernels_vs_reference_closures(gpu, path, dtype):
    """pack_fwd / unpack_fwd / pack_rev / unpack_rev, each launched explicitly (no direct-mode
    short cut), messages carried between the simulated ranks through a host mailbox."""
    torch = gpu
    scat = pkg("scatterer")
    d = np.load(path)
    meshes, od, gd = _partition(d)
    R = len(meshes)
    k = scat.HipHaloKernels(torch.float64 if dtype == np.float64 else torch.float32)
    dev = torch.device("cuda", 0)
    for kind in ("fwd", "rev"):
        bufs = [torch.from_numpy(d[f"in_{r}"].astype(dtype)).to(dev) for r in range(R)]
        box = {}
        for r in range(R):  # pack + "send"
            N = meshes[r].nlocal
            side = gd[r] if kind == "fwd" else od[r]
            idx, size, off, ranks = side
            send = k.buffer(len(idx))
            if len(idx):
                if kind == "fwd":
                    k.pack_fwd(bufs[r], send, k.index_tensor(idx))
                else:
                    k.pack_rev(bufs[r], send, k.index_tensor(idx), N)
            host = send.cpu().numpy()
            for i, dst in enumerate(ranks):
                box[(r, int(dst))] = host[off[i]:off[i + 1]]
        for r in range(R):  # "recv" + unpack
            N = meshes[r].nlocal
            side = od[r] if kind == "fwd" else gd[r]
            idx, size, off, ranks = side
            if len(idx) == 0:
                continue
            recv = torch.from_numpy(np.concatenate([box[(int(src), r)] for src in ranks])).to(dev)
            if kind == "fwd":
                k.unpack_fwd(recv, bufs[r], k.index_tensor(idx), N)
            else:
                k.unpack_rev(recv, bufs[r], k.index_tensor(idx))
        torch.cuda.synchronize()
        for r in range(R):
            ref = d[f"ref_{kind}_{r}"]
            got = bufs[r].cpu().numpy()
            if kind == "fwd":  # pure copies: exact
                assert np.array_equal(got, ref.astype(dtype)), f"{kind} rank {r}"
            else:
                assert np.allclose(got, ref, rtol=0, atol=1e-13 if dtype == np.float64 else 2e-6), f"{kind} rank {r}"


TRANSPORTS = ["local", "peer"]  # in-process worlds: stream-ordered copies | the PEER protocol on plain pointers


def _local_comm(scat, wid, R, r, transport):
    return scat.NativeComm(local=(wid, R, r), transport="peer" if transport == "peer" else "rccl")


def _native_scatterers(meshes, od, gd, float_type, transport="local"):
    scat = pkg("scatterer")
    wid = next(_world_ids)
    R = len(meshes)
    comms = [_local_comm(scat, wid, R, r, transport) for r in range(R)]
    fwd = [scat.scatter_forward(comms[r], od[r], gd[r], meshes[r].nlocal, float_type) for r in range(R)]
    rev = [scat.scatter_reverse(comms[r], od[r], gd[r], meshes[r].nlocal, float_type) for r in range(R)]
    return comms, fwd, rev


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("path", golden_files("scatter_"), ids=lambda p: p.split("/")[-1][:-4])
def test_native_halo_vs_reference_closures(gpu, path, dtype, transport):
    """fus_halo_forward / fus_halo_reverse (C ABI) on the reference's fixtures; BoxMesh numbers its
    ghosts owner by owner, so this is the direct mode (ghost block = message buffer)."""
    torch = gpu
    d = np.load(path)
    meshes, od, gd = _partition(d)
    R = len(meshes)
    comms, fwd, rev = _native_scatterers(meshes, od, gd, dtype, transport)
    assert all(f.direct for f, m in zip(fwd, meshes) if m.nghost > 0)
    dev = torch.device("cuda", 0)
    for kind, closures in (("fwd", fwd), ("rev", rev)):
        bufs = [torch.from_numpy(d[f"in_{r}"].astype(dtype)).to(dev) for r in range(R)]
        for rep in range(2 if kind == "fwd" else 1):  # forward is idempotent: run it twice (buffer reuse)
            for r in range(R):
                closures[r].begin(bufs[r])
            for r in range(R):
                closures[r].end(bufs[r])
        torch.cuda.synchronize()
        for r in range(R):
            ref, got = d[f"ref_{kind}_{r}"], bufs[r].cpu().numpy()
            if kind == "fwd":
                assert np.array_equal(got, ref.astype(dtype))
            else:
                assert np.allclose(got, ref, rtol=0, atol=1e-13 if dtype == np.float64 else 2e-6)
    assert all(sc.status()["timeouts"] == 0 for sc in fwd + rev)


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("ghost_order", ["lex", 3, 11])
@pytest.mark.parametrize("P,shape,grid", [(2, (4, 4, 2), (2, 2, 1)), (3, (4, 4, 4), (2, 2, 2)), (2, (6, 2, 2), (3, 1, 1))])
def test_native_halo_arbitrary_ghost_numbering(gpu, P, shape, grid, ghost_order, transport):
    """Ghosts not grouped by owner (every real dolfinx IndexMap): the exchange must go through
    unpack_fwd / pack_rev.  Checked against the numpy restatement of the reference's closures
    (oracle/oracle_np.py, itself pinned by the golden fixtures) and by the owner-value property."""
    from oracle import oracle_np

    torch = gpu
    boxmesh, utils = pkg("boxmesh"), pkg("utils")
    R = int(np.prod(grid))
    meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r, ghost_order=ghost_order) for r in range(R)]
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    comms, fwd, rev = _native_scatterers(meshes, od, gd, np.float64, transport)
    if R > 3 and ghost_order != "lex":
        assert not all(f.direct for f in fwd), "the test must exercise the non-direct path"
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(4)
    host = [rng.standard_normal(m.ndofs) for m in meshes]
    nl = [m.nlocal for m in meshes]
    for kind, closures, oracle in (("fwd", fwd, oracle_np.scatter_forward_all), ("rev", rev, oracle_np.scatter_reverse_all)):
        ref = [h.copy() for h in host]
        oracle(ref, od, gd, nl)
        bufs = [torch.from_numpy(h).to(dev) for h in host]
        for r in range(R):
            closures[r].begin(bufs[r])
        for r in range(R):
            closures[r].end(bufs[r])
        torch.cuda.synchronize()
        for r in range(R):
            got = bufs[r].cpu().numpy()
            assert np.allclose(got, ref[r], rtol=0, atol=1e-13), f"{kind} rank {r}"
    # property: after a forward scatter of the global lexicographic id every ghost holds its own id
    bufs = []
    for r, m in enumerate(meshes):
        lex = m.global_lexicographic_ids().astype(np.float64)
        v = lex.copy()
        v[m.nlocal:] = -1.0
        bufs.append((torch.from_numpy(v).to(dev), lex))
    for r in range(R):
        fwd[r].begin(bufs[r][0])
    for r in range(R):
        fwd[r].end(bufs[r][0])
    torch.cuda.synchronize()
    for t, lex in bufs:
        assert np.array_equal(t.cpu().numpy(), lex)
    assert all(sc.status()["timeouts"] == 0 for sc in fwd + rev)


@pytest.mark.parametrize("transport,overlap", [("local", "split"), ("local", "concurrent"), ("local", False),
                                               ("peer", "concurrent"), ("peer", "split"), ("peer", False)],
                         ids=["local-split", "local-concurrent", "local-sequential", "peer-concurrent", "peer-split", "peer-sequential"])
@pytest.mark.parametrize("P,cells,grid,ghost_order", [
    (4, (4, 4, 4), (2, 1, 1), "owner"),
    (3, (4, 4, 2), (2, 2, 1), 7),
    (2, (4, 4, 4), (2, 2, 2), 7),
    (4, (6, 4, 4), (2, 2, 1), "lex"),
], ids=["2ranks-direct", "4ranks-permuted", "8ranks-permuted", "4ranks-lex"])
def test_partitioned_apply_async_transport_one_gpu(gpu, oracle_c, P, cells, grid, ghost_order, transport, overlap):
    """HaloApply with the exchange issued from C++ on its own stream, all ranks on cuda:0 in this
    process, NO host synchronisation between pack, exchange, unpack and the operator kernels: the
    begin | interior | end -> boundary -> begin | interior | end schedule runs with real
    asynchrony.  Must equal the serial apply dof for dof.  Schedules: "split" (interior half |
    boundary | interior half on one stream, lead slices), "concurrent" (one interior launch; boundary cells and both
    exchanges on a side stream next to it), sequential."""
    torch = gpu
    schedule = overlap if overlap else "split"
    overlap = bool(overlap)
    boxmesh, scat, ops, gll, pre = (pkg(m) for m in ("boxmesh", "scatterer", "operators", "gll", "precompute"))
    R = int(np.prod(grid))
    wid = next(_world_ids)
    dev = torch.device("cuda", 0)
    pts, wts, D = gll.tabulate_1d(P)
    n = P + 1
    w3 = gll.tensor_weights_3d(wts)
    dg = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))
    op = ops.stiffness_operator(P, D.flatten(), np.float64)
    ranks = []
    for r in range(R):
        mesh = boxmesh.BoxMesh(P, cells, grid=grid, rank=r, perturb=0.16, seed=3, ghost_order=ghost_order)
        G = np.zeros((mesh.ncells, n**3, 6))
        pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), mesh.ncells, dg, w3)
        x = ref_field(mesh.dof_coordinates())
        x[mesh.nlocal:] = -777.0  # ghosts are stale until the forward scatter
        ranks.append(dict(mesh=mesh, x=torch.from_numpy(x).to(dev), y=torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev),
                          cc=torch.from_numpy(global_cell_constants(mesh)).to(dev), G=torch.from_numpy(G).to(dev),
                          dm=torch.from_numpy(mesh.dofmap).to(dev)))

    utils = pkg("utils")  # an in-process world: every rank's halo plan at once
    od, gd = utils.compute_scatterer_data_all([rk["mesh"].index_map for rk in ranks])
    halos = []
    for r, rk in enumerate(ranks):
        comm = _local_comm(scat, wid, R, r, transport)
        # lead slices (the small launches that open each overlapped region) forced on every other rank
        halos.append(scat.HaloApply(rk["mesh"], op, comm, np.float64, overlap=overlap, plan=(od[r], gd[r]),
                                    lead_cells=(2 if r % 2 == 0 else 0), schedule=schedule))
        split = overlap and schedule == "split"
        assert halos[-1].schedule_kind == (schedule if overlap else "sequential")
        assert halos[-1].lead_cells in ((0, 1, 2) if (r % 2 == 0 and split) else (0,))  # clamped to half the interior
    assert halos[0].lead_cells == (2 if split else 0)  # rank 0 of these partitions has >= 4 interior cells
    for rep in range(2):  # second apply: buffers / events reused while the first may still be in flight
        for rk in ranks:
            rk["y"].zero_()
        gens = [h.apply_schedule(rk["x"], rk["cc"], rk["y"], rk["G"], rk["dm"]) for h, rk in zip(halos, ranks)]
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
    torch.cuda.synchronize()
    # serial reference
    pb = build_problem(P, cells, perturb=0.16, seed=3)
    ms = pb["mesh"]
    y_ser = np.zeros(ms.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], global_cell_constants(ms), y_ser, pb["G"], ms.dofmap)
    seen = np.zeros(ms.ndofs, dtype=int)
    for rk in ranks:
        m = rk["mesh"]
        lex = m.global_lexicographic_ids()
        seen[lex[: m.nlocal]] += 1
        assert rel_l2(rk["y"].cpu().numpy()[: m.nlocal], y_ser[lex[: m.nlocal]]) < 1e-12
        assert np.allclose(rk["x"].cpu().numpy(), pb["x"][lex], rtol=0, atol=1e-12)  # ghosts refreshed
    assert np.all(seen == 1)
    assert all(h.health() == 0 for h in halos)


@pytest.mark.parametrize("transport,overlap", [("peer", "concurrent"), ("peer", "split"), ("local", "concurrent"), ("local", False)],
                         ids=["peer-concurrent", "peer-split", "local-concurrent", "local-sequential"])
@pytest.mark.parametrize("P,cells,grid,ghost_order", [
    (4, (12, 7, 7), (2, 1, 1), "owner"),
    (3, (16, 16, 9), (2, 2, 1), 7),
    (4, (14, 14, 14), (2, 2, 2), 5),
], ids=["2ranks-direct", "4ranks-permuted", "8ranks-permuted"])
def test_partitioned_mass_apply_row_split_one_gpu(gpu, oracle_c, P, cells, grid, ghost_order, transport, overlap):
    """the partitioned MASS apply on the atomic-free kernel.  ``HaloApply`` splits a row-wise operator by DOF,
    not by cell: set A (owned dofs the reverse exchange does not add into) in one launch next to the exchanges, set B (ghost dofs
    and owned dofs ghosted elsewhere) between them -- so no launch and no receive kernel adds into one y[d] concurrently.  2 / 4 / 8
    in-process ranks, every rank's owned part == the serial oracle's cell mass apply (numba-cpu/operators.py:19-68), twice in a row;
    the kernel really is the gather kernel (>= 32 768 entries per rank)."""
    torch = gpu
    schedule = overlap if overlap else "split"
    overlap = bool(overlap)
    boxmesh, scat, ops, gll, pre, utils = (pkg(m) for m in ("boxmesh", "scatterer", "operators", "gll", "precompute", "utils"))
    R = int(np.prod(grid))
    wid = next(_world_ids)
    dev = torch.device("cuda", 0)
    pts, wts, D = gll.tabulate_1d(P)
    n = P + 1
    w3 = gll.tensor_weights_3d(wts)
    dg = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))
    op = ops.mass_operator(n**3, np.float64)
    ranks = []
    for r in range(R):
        mesh = boxmesh.BoxMesh(P, cells, grid=grid, rank=r, perturb=0.16, seed=3, ghost_order=ghost_order)
        assert mesh.ncells * n**3 >= ops._MASS_PLAN_MIN_ENTRIES
        detJ = np.zeros((mesh.ncells, n**3))
        pre.compute_scaled_jacobian_determinant(detJ, (mesh.x_dofs, mesh.x_g), mesh.ncells, dg, w3)
        x = ref_field(mesh.dof_coordinates())
        x[mesh.nlocal:] = -777.0  # ghosts are stale until the forward scatter
        ranks.append(dict(mesh=mesh, x=torch.from_numpy(x).to(dev), y=torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev),
                          cc=torch.from_numpy(global_cell_constants(mesh)).to(dev), detJ=torch.from_numpy(detJ).to(dev),
                          dm=torch.from_numpy(mesh.dofmap).to(dev)))
    od, gd = utils.compute_scatterer_data_all([rk["mesh"].index_map for rk in ranks])
    halos = [scat.HaloApply(rk["mesh"], op, _local_comm(scat, wid, R, r, transport), np.float64, overlap=overlap, plan=(od[r], gd[r]),
                            schedule=schedule) for r, rk in enumerate(ranks)]
    for h, rk in zip(halos, ranks):
        marks = h.row_split(rk["dm"], rk["mesh"].ndofs)
        assert marks is not None and marks.dtype == torch.uint8  # the row split is what this apply uses
        nl = rk["mesh"].nlocal
        assert bool((marks[nl:] == 1).all()) and int(marks[:nl].sum().item()) == np.unique(np.asarray(gd[halos.index(h)][0])).size
    for rep in range(2):
        for rk in ranks:
            rk["y"].zero_()
        live = [h.apply_schedule(rk["x"], rk["cc"], rk["y"], rk["detJ"], rk["dm"]) for h, rk in zip(halos, ranks)]
        while live:
            nxt = []
            for g in live:
                try:
                    next(g)
                    nxt.append(g)
                except StopIteration:
                    pass
            live = nxt
    torch.cuda.synchronize()
    ms = boxmesh.BoxMesh(P, cells, perturb=0.16, seed=3)
    detJ_s = np.zeros((ms.ncells, n**3))
    pre.compute_scaled_jacobian_determinant(detJ_s, (ms.x_dofs, ms.x_g), ms.ncells, dg, w3)
    x_s = ref_field(ms.dof_coordinates())
    y_ser = np.zeros(ms.ndofs)
    oracle_c.mass_apply(x_s, global_cell_constants(ms), y_ser, detJ_s, ms.dofmap)
    seen = np.zeros(ms.ndofs, dtype=int)
    for rk in ranks:
        m = rk["mesh"]
        lex = m.global_lexicographic_ids()
        seen[lex[: m.nlocal]] += 1
        assert rel_l2(rk["y"].cpu().numpy()[: m.nlocal], y_ser[lex[: m.nlocal]]) < 1e-13
        assert np.allclose(rk["x"].cpu().numpy(), x_s[lex], rtol=0, atol=1e-12)  # ghosts refreshed
    assert np.all(seen == 1) and all(h.health() == 0 for h in halos)
    # the launch schedule without exchange and the plain sequence of launches add the same local contributions
    rk, h = ranks[0], halos[0]
    ya, yb = torch.zeros_like(rk["y"]), torch.zeros_like(rk["y"])
    h.apply_no_exchange(rk["x"], rk["cc"], ya, rk["detJ"], rk["dm"])
    h.apply_local_only(rk["x"], rk["cc"], yb, rk["detJ"], rk["dm"])
    torch.cuda.synchronize()
    assert torch.equal(ya, yb)  # the atomic-free kernel is bitwise reproducible


def _run_peer_world(world, mode, args, timeout=300):
    """``world`` real processes sharing cuda:0, halo transport PEER over HIP IPC handles (tests/_peer_worker.py)."""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "_peer_worker.py"), mode, str(r), str(world), str(port)]
                              + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    bad = [r for r, (p, out) in enumerate(zip(procs, outs)) if p.returncode != 0 or f"PEER_WORKER_OK {r}" not in out]
    assert not bad, "\n".join(f"---- process {r} (exit {procs[r].returncode}):\n{outs[r][-1500:]}" for r in bad)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("fixture", ["scatter_P2_4x2x2_grid2x1x1", "scatter_P3_2x4x2_grid1x2x1", "scatter_P2_4x4x2_grid2x2x1"])
def test_peer_transport_real_processes_vs_reference_closures(gpu, fixture, dtype):
    """2 and 4 REAL processes sharing cuda:0; arenas mapped with hipIpcOpenMemHandle; the reference closures' own
    inputs and outputs (tests/golden/scatter_*.npz), then 50 exchanges back to back."""
    import os

    path = os.path.join(os.path.dirname(__file__), "golden", fixture + ".npz")
    d = np.load(path)
    _run_peer_world(int(np.prod(d["grid"])), "golden", [path, dtype])


@pytest.mark.parametrize("P,cells,grid,ghost_order,schedule", [
    (4, (4, 4, 4), (2, 1, 1), "owner", "concurrent"),
    (3, (4, 4, 2), (2, 2, 1), 7, "concurrent"),
    (2, (6, 4, 4), (2, 2, 1), "lex", "split"),
], ids=["2procs-direct", "4procs-permuted", "4procs-lex-split"])
def test_peer_transport_real_processes_partitioned_apply(gpu, P, cells, grid, ghost_order, schedule):
    """HaloApply over the PEER transport with one PROCESS per rank (index exchange over gloo, arenas over HIP IPC):
    every rank's owned part of y == the serial C oracle's, ghosts of x refreshed."""
    _run_peer_world(int(np.prod(grid)), "apply", [P, *cells, *grid, ghost_order, schedule])


def _run_mpi_world(world, mode, args, tmp_path, timeout=300):
    """``world`` real processes sharing cuda:0 whose ONLY communicator is an MPI-style one (tests/fake_mpi.py over ``tmp_path``):
    no torch.distributed group anywhere (tests/_mpi_worker.py asserts it)."""
    import os
    import subprocess
    import sys

    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "_mpi_worker.py"), mode, str(r), str(world), str(tmp_path)]
                              + [str(a) for a in args], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    bad = [r for r, (p, out) in enumerate(zip(procs, outs)) if p.returncode != 0 or f"MPI_WORKER_OK {r}" not in out]
    assert not bad, "\n".join(f"---- process {r} (exit {procs[r].returncode}):\n{outs[r][-1500:]}" for r in bad)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("fixture", ["scatter_P2_4x2x2_grid2x1x1", "scatter_P2_4x4x2_grid2x2x1"])
def test_mpi_communicator_real_processes_vs_reference_closures(gpu, tmp_path, fixture, dtype):
    """the reference's own lines -- ``compute_scatterer_data(index_map)``, ``scatter_reverse / scatter_forward(comm,
    owners_data, ghosts_data, N, float_type)`` with ``comm`` an MPI communicator (cuda/demo_linear_box.py:41, 192, 206-207) -- on 2 and 4
    real processes sharing cuda:0, PEER transport bootstrapped over that communicator alone, against the reference closures'
    outputs (tests/golden/scatter_*.npz)."""
    import os

    path = os.path.join(os.path.dirname(__file__), "golden", fixture + ".npz")
    d = np.load(path)
    _run_mpi_world(int(np.prod(d["grid"])), "golden", [path, dtype], tmp_path)


@pytest.mark.parametrize("fused", [0, 1], ids=["reference-sequence", "fused"])
def test_mpi_communicator_real_processes_rk4_solver(gpu, tmp_path, fused):
    """``LinearSpectral3D(mesh, comm=<MPI communicator>)`` on two real processes == the reference-driven loop's fixture."""
    import os

    _run_mpi_world(2, "solver", [os.path.join(os.path.dirname(__file__), "golden", "rk4_P2_4x2x2_pert_2ranks.npz"), fused], tmp_path)


def test_scatter_rejects_what_is_not_a_communicator(gpu, tmp_path):
    _run_mpi_world(1, "badcomm", [], tmp_path)


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("seed,permute", [(0, False), (1, True), (2, True)])
def test_random_plans_uneven_multi_chunk_segments(gpu, seed, permute, transport):
    """Halo plans that no box partition produces: 4 ranks, every ordered pair with its own message size from
    {0, 1, 700, 1024, 1025, 2500, 5000} (segments of several 1 024-element chunks next to empty and one-element ones, so the
    per-segment completion counters, credits and chunk tables of the PEER kernels all see uneven shapes), an owned dof
    ghosted by several ranks (reverse adds from two neighbours land on one entry), ghosts grouped by owner or permuted.
    Forward and reverse, three exchanges each, against numpy."""
    torch = gpu
    scat = pkg("scatterer")
    rng = np.random.default_rng(100 + seed)
    R, nlocal = 4, 20000
    sizes = [0, 1, 700, 1024, 1025, 2500, 5000]
    count = {(q, r): (int(rng.choice(sizes)) if q != r else 0) for q in range(R) for r in range(R)}  # q ghosts count dofs of r
    owned_pick = {(q, r): rng.choice(nlocal, size=count[(q, r)], replace=False).astype(np.int64) for q in range(R) for r in range(R)}
    od, gd, nghost = [], [], []
    ghost_pos = {}
    for q in range(R):  # owners side of q: its ghost block = the owners' segments in ascending owner order
        owners = [r for r in range(R) if count[(q, r)] > 0]
        ng = sum(count[(q, r)] for r in owners)
        pos = rng.permutation(ng).astype(np.int64) if permute else np.arange(ng, dtype=np.int64)
        off = np.concatenate(([0], np.cumsum([count[(q, r)] for r in owners]))).astype(np.int64)
        for i, r in enumerate(owners):
            ghost_pos[(q, r)] = pos[off[i]: off[i + 1]]
        od.append([pos, np.array([count[(q, r)] for r in owners], dtype=np.int64), off, np.array(owners, dtype=np.int32)])
        nghost.append(ng)
    for r in range(R):  # ghosts side of r: its owned dofs in the order each ghosting rank packs them
        ghosters = [q for q in range(R) if count[(q, r)] > 0]
        idx = np.concatenate([owned_pick[(q, r)] for q in ghosters]) if ghosters else np.zeros(0, np.int64)
        off = np.concatenate(([0], np.cumsum([count[(q, r)] for q in ghosters]))).astype(np.int64)
        gd.append([idx.astype(np.int64), np.array([count[(q, r)] for q in ghosters], dtype=np.int64), off, np.array(ghosters, dtype=np.int32)])
    wid = next(_world_ids)
    comms = [_local_comm(scat, wid, R, r, transport) for r in range(R)]
    fwd = [scat.scatter_forward(comms[r], od[r], gd[r], nlocal, np.float64) for r in range(R)]
    rev = [scat.scatter_reverse(comms[r], od[r], gd[r], nlocal, np.float64) for r in range(R)]
    dev = torch.device("cuda", 0)
    for rep in range(3):
        host = [rng.standard_normal(nlocal + nghost[r]) for r in range(R)]
        # forward
        bufs = [torch.from_numpy(h).to(dev) for h in host]
        for r in range(R):
            fwd[r].begin(bufs[r])
        for r in range(R):
            fwd[r].end(bufs[r])
        torch.cuda.synchronize()
        for q in range(R):
            ref = host[q].copy()
            for r in range(R):
                if count[(q, r)]:
                    ref[nlocal + ghost_pos[(q, r)]] = host[r][owned_pick[(q, r)]]
            assert np.array_equal(bufs[q].cpu().numpy(), ref), f"forward, rank {q}, rep {rep}"
        # reverse
        bufs = [torch.from_numpy(h).to(dev) for h in host]
        for r in range(R):
            rev[r].begin(bufs[r])
        for r in range(R):
            rev[r].end(bufs[r])
        torch.cuda.synchronize()
        for r in range(R):
            ref = host[r].copy()
            for q in range(R):
                if count[(q, r)]:
                    np.add.at(ref, owned_pick[(q, r)], host[q][nlocal + ghost_pos[(q, r)]])
            assert np.allclose(bufs[r].cpu().numpy(), ref, rtol=0, atol=1e-13), f"reverse, rank {r}, rep {rep}"
    assert all(sc.status()["timeouts"] == 0 for sc in fwd + rev)


@pytest.mark.parametrize("fused", [0, 1], ids=["reference-sequence", "fused"])
def test_peer_transport_real_processes_rk4_solver_vs_reference_driven_loop(gpu, fused):
    """Two real processes, PEER transport, the whole linear RK4 solver (set-up reverse scatter of the lumped mass, per
    stage the grouped forward scatter of (u_n, v_n) and the reverse scatter of b, concurrent schedule) against the
    fixture produced by the reference's own operators and scatter closures in the same loop."""
    import os

    _run_peer_world(2, "solver", [os.path.join(os.path.dirname(__file__), "golden", "rk4_P2_4x2x2_pert_2ranks.npz"), fused])


def test_peer_transport_real_processes_soak_with_host_skew(gpu):
    """Four real processes, 1 500 forward + 1 500 reverse exchanges back to back with data that changes every iteration
    and random host-side delays on every rank (a rank posts up to 4 ms late now and then): sequence flags and credits
    must keep every message apart -- every ghost value and every owned sum is checked exactly on the device each time."""
    _run_peer_world(4, "soak", [1500], timeout=600)


def test_peer_transport_dead_peer_times_out_instead_of_hanging(gpu):
    """Every device-side wait of the PEER transport is bounded: a rank whose neighbour never sends gets its time-out
    counted (``status()["timeouts"]``) after FUS_IPC_SPIN_SECONDS and its kernels drain; later exchanges of that halo do
    not wait at all."""
    _run_peer_world(2, "deadpeer", [])


def test_peer_transport_solver_raises_when_a_neighbour_stops(gpu):
    """A timed-out exchange must be LOUD (ADVICE medium): rank 1 leaves the time loop after two steps;
    rank 0's ``LinearSpectral3D.rk4`` raises instead of handing back a pressure field computed from stale ghosts, and the
    failure reaches rank 1 through poisoned flags: its next ``rk4`` raises too, with no time-out of its own."""
    _run_peer_world(2, "solver_deadpeer", [], timeout=300)


@pytest.mark.parametrize("what", ["apply", "solver"])
@pytest.mark.parametrize("P,cells,grid,ghost_order,rpp", [
    (4, (4, 4, 4), (2, 2, 2), 7, 2),
    (2, (6, 4, 4), (2, 2, 2), "owner", 2),
], ids=["P4-permuted", "P2-direct-uneven"])
def test_peer_transport_8_ranks_2x2x2_on_4_processes(gpu, P, cells, grid, ghost_order, rpp, what):
    """The exact 8-rank configuration of BASELINE config 4 (2x2x2 blocks: rank 0 ghosted by 7 ranks, rank 7 ghosting from 7
    owners; face, (P n + 1)-element edge and 1-element corner segments) over the PEER transport with REAL processes.  The
    pool's process guard allows at most 6 processes on the card, so the 8 ranks run on 4 processes of 2 ranks each:
    neighbours of the same process are reached through plain pointers, the others through HIP IPC mappings, by the same
    kernels.  apply: HaloApply (concurrent schedule, fork / join folded into the exchange kernels) == the serial C oracle;
    solver: 5 steps of the fused linear solver on the partition == the one-rank solver.  (More hardware queues than the
    card has: the processes' queues are time-sliced, so the device-side waits get 60 s here.)"""
    R = int(np.prod(grid))
    _run_peer_world(R // rpp, "hybrid", [P, *cells, *grid, ghost_order, rpp, what], timeout=600)


@pytest.mark.parametrize("P,cells,grid,ghost_order,rpp,geom", [
    (3, (4, 4, 4), (2, 2, 2), 7, 2, ""),
    (3, (4, 4, 4), (2, 2, 2), "owner", 2, "geom"),
    (4, (4, 3, 3), (2, 1, 1), 5, 1, ""),
], ids=["8ranks-4procs-general-G", "8ranks-4procs-in-kernel-geometry", "2procs-general-G"])
def test_peer_transport_westervelt_solver_real_processes(gpu, P, cells, grid, ghost_order, rpp, geom):
    """BASELINE config 5's loop (fused Westervelt solver: set-up scatter of three assembled diagonals, grouped forward scatter
    of (w, v_n) and reverse of b per stage, concurrent schedule) over the PEER transport with REAL processes: the 2x2x2 world
    as 4 processes of 2 ranks, and 2 processes of one rank each; 4 steps == the one-rank solver (itself pinned to the
    reference-driven loop, tests/test_rk4_golden.py)."""
    R = int(np.prod(grid))
    _run_peer_world(R // rpp, "hybrid", [P, *cells, *grid, ghost_order, rpp, "westervelt"] + ([geom] if geom else []), timeout=600)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("direct", [True, False], ids=["direct", "permuted"])
@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_rccl_self_exchange(gpu, dtype, direct, transport):
    """The RCCL transport (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd issued by libfusgpu.so
    on its own stream) in a 1-rank world whose only rank is its own neighbour; and the PEER transport the same way
    (the rank's own arena is its neighbour's: several chunks per segment, credits, sequence flags)."""
    torch = gpu
    scat = pkg("scatterer")
    comm = scat.NativeComm(transport=transport)  # no process group: world size 1, unique id stays local
    assert comm.size == 1 and comm.backend == transport
    rng = np.random.default_rng(12)
    N, ng = 5000, 2700
    o_idx = np.arange(ng) if direct else rng.permutation(ng)
    g_idx = rng.choice(N, size=ng, replace=False)
    od = [o_idx.astype(np.int64), np.array([ng]), np.array([0, ng]), np.array([0], dtype=np.int32)]
    gd = [g_idx.astype(np.int64), np.array([ng]), np.array([0, ng]), np.array([0], dtype=np.int32)]
    fwd = scat.scatter_forward(comm, od, gd, N, dtype)
    rev = scat.scatter_reverse(comm, od, gd, N, dtype)
    assert fwd.direct == direct
    host = rng.standard_normal(N + ng).astype(dtype)
    dev = torch.device("cuda", 0)
    buf = torch.from_numpy(host).to(dev)
    fwd(buf)
    ref = host.copy()
    ref[N + o_idx] = host[g_idx]
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy(), ref)
    buf = torch.from_numpy(host).to(dev)
    rev(buf)
    ref = host.copy()
    np.add.at(ref, g_idx, host[N + o_idx])
    torch.cuda.synchronize()
    assert np.allclose(buf.cpu().numpy(), ref, rtol=0, atol=1e-14 if dtype == np.float64 else 1e-6)
    for _ in range(20):  # buffer and flag reuse
        fwd(buf)
    torch.cuda.synchronize()
    assert fwd.status()["timeouts"] == 0 and rev.status()["timeouts"] == 0
    fwd.close(), rev.close(), comm.close()
    assert comm.handle is None  # the communicator was destroyed (it refuses while halo objects are alive)


def test_halo_create_rejects_bad_plans(gpu):
    """fus_halo_create range-checks what the pack / unpack kernels would index with (the reference does not):
    ghost positions outside the ghost block, owned indices outside [0, nlocal), neighbour ranks outside the
    communicator, sizes that do not add up -- FUS_ERR_INVALID_ARGUMENT, nothing launched."""
    import ctypes as C

    scat, lib = pkg("scatterer"), pkg("_lib").load()
    comm = scat.NativeComm(local=(next(_world_ids), 2, 0))

    def create(nlocal, nghost, o_ranks, o_sizes, o_idx, g_ranks, g_sizes, g_idx):
        arr = lambda a, dt: np.ascontiguousarray(a, dtype=dt)  # noqa: E731
        keep = [arr(o_ranks, np.int32), arr(o_sizes, np.int64), arr(o_idx, np.int64), arr(g_ranks, np.int32), arr(g_sizes, np.int64), arr(g_idx, np.int64)]
        h = C.c_void_p()
        rc = lib.fus_halo_create(comm.handle, 8, nlocal, nghost, len(o_ranks), *(k.ctypes.data_as(C.c_void_p) for k in keep[:3]),
                                 len(g_ranks), *(k.ctypes.data_as(C.c_void_p) for k in keep[3:]), C.byref(h))
        if rc == 0:
            lib.fus_halo_destroy(h)
        return rc

    ok = create(10, 3, [1], [3], [0, 1, 2], [1], [2], [4, 9])
    assert ok == 0
    assert create(10, 3, [1], [3], [0, 1, 3], [1], [2], [4, 9]) == -1     # ghost position 3 outside the ghost block
    assert create(10, 3, [1], [3], [0, 1, 2], [1], [2], [4, 10]) == -1    # owned index 10 outside [0, nlocal)
    assert create(10, 3, [2], [3], [0, 1, 2], [1], [2], [4, 9]) == -1     # rank 2 in a 2-rank world
    assert create(10, 2, [1], [3], [0, 1, 2], [1], [2], [4, 9]) == -1     # more owner entries than ghosts
    assert create(10, 3, [1], [-1], [0, 1, 2], [1], [2], [4, 9]) == -1    # negative size
    assert lib.fus_halo_create(None, 8, 10, 3, 0, None, None, None, 0, None, None, None, C.byref(C.c_void_p())) == -1
    comm.close()


def test_grouped_exchange_two_vectors_rccl_self(gpu):
    """fus_halo_forward_begin_group / reverse: two vectors in ONE RCCL group (two messages to the same peer,
    matched in issue order) in a 1-rank world that is its own neighbour; one direct halo, one permuted."""
    torch = gpu
    scat = pkg("scatterer")
    comm = scat.NativeComm()
    rng = np.random.default_rng(21)
    N, ng = 3000, 700
    g_idx = rng.choice(N, size=ng, replace=False)
    mk = lambda o: ([o.astype(np.int64), np.array([ng]), np.array([0, ng]), np.array([0], dtype=np.int32)],  # noqa: E731
                    [g_idx.astype(np.int64), np.array([ng]), np.array([0, ng]), np.array([0], dtype=np.int32)])
    o_a, o_b = np.arange(ng), rng.permutation(ng)
    fa, fb = scat.scatter_forward(comm, *mk(o_a), N, np.float64), scat.scatter_forward(comm, *mk(o_b), N, np.float64)
    ra, rb = scat.scatter_reverse(comm, *mk(o_a), N, np.float64), scat.scatter_reverse(comm, *mk(o_b), N, np.float64)
    dev = torch.device("cuda", 0)
    ha, hb = rng.standard_normal(N + ng), rng.standard_normal(N + ng)
    a, b = torch.from_numpy(ha).to(dev), torch.from_numpy(hb).to(dev)
    for sc, vec, wk in scat.begin_all([(fa, a), (fb, b)]):
        sc.end(vec, wk)
    torch.cuda.synchronize()
    ea, eb = ha.copy(), hb.copy()
    ea[N + o_a] = ha[g_idx]
    eb[N + o_b] = hb[g_idx]
    assert np.array_equal(a.cpu().numpy(), ea) and np.array_equal(b.cpu().numpy(), eb)
    a, b = torch.from_numpy(ha).to(dev), torch.from_numpy(hb).to(dev)
    for sc, vec, wk in scat.begin_all([(ra, a), (rb, b)]):
        sc.end(vec, wk)
    torch.cuda.synchronize()
    ea, eb = ha.copy(), hb.copy()
    np.add.at(ea, g_idx, ha[N + o_a])
    np.add.at(eb, g_idx, hb[N + o_b])
    assert np.allclose(a.cpu().numpy(), ea, rtol=0, atol=1e-14) and np.allclose(b.cpu().numpy(), eb, rtol=0, atol=1e-14)


@pytest.mark.parametrize("transport", TRANSPORTS)
@pytest.mark.parametrize("ghost_order", ["owner", 9])
def test_grouped_exchange_two_vectors_in_process_ranks(gpu, ghost_order, transport):
    """The same through the in-process transport with 4 ranks: u and v forward-scattered as one unit per
    rank (what an RK4 stage does), then both reverse-scattered as one unit."""
    from oracle import oracle_np

    torch = gpu
    scat, boxmesh, utils = pkg("scatterer"), pkg("boxmesh"), pkg("utils")
    grid, R = (2, 2, 1), 4
    meshes = [boxmesh.BoxMesh(2, (4, 4, 2), grid=grid, rank=r, ghost_order=ghost_order) for r in range(R)]
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    wid = next(_world_ids)
    comms = [_local_comm(scat, wid, R, r, transport) for r in range(R)]
    mk = lambda fn: [fn(comms[r], od[r], gd[r], meshes[r].nlocal, np.float64) for r in range(R)]  # noqa: E731
    fu, fv, ru, rv = mk(scat.scatter_forward), mk(scat.scatter_forward), mk(scat.scatter_reverse), mk(scat.scatter_reverse)
    rng = np.random.default_rng(6)
    hu, hv = [rng.standard_normal(m.ndofs) for m in meshes], [rng.standard_normal(m.ndofs) for m in meshes]
    nl = [m.nlocal for m in meshes]
    dev = torch.device("cuda", 0)
    for closures, oracle in (((fu, fv), oracle_np.scatter_forward_all), ((ru, rv), oracle_np.scatter_reverse_all)):
        eu, ev = [h.copy() for h in hu], [h.copy() for h in hv]
        oracle(eu, od, gd, nl)
        oracle(ev, od, gd, nl)
        du, dv = [torch.from_numpy(h).to(dev) for h in hu], [torch.from_numpy(h).to(dev) for h in hv]
        pending = [scat.begin_all([(closures[0][r], du[r]), (closures[1][r], dv[r])]) for r in range(R)]
        for p in pending:
            for sc, vec, wk in p:
                sc.end(vec, wk)
        torch.cuda.synchronize()
        for r in range(R):
            assert np.allclose(du[r].cpu().numpy(), eu[r], rtol=0, atol=1e-13)
            assert np.allclose(dv[r].cpu().numpy(), ev[r], rtol=0, atol=1e-13)


def test_peer_failed_exchange_poisons_the_neighbour(gpu, monkeypatch):
    """In-process world of two ranks.  The ghosting rank posts its forward exchange alone: its receive gives up after
    FUS_IPC_SPIN_SECONDS (time-out, halo dead) and the credit it returns is POISONED.  When the owner posts later, its send
    reads the poisoned credit: it counts it, does not deliver, dies too -- ``health()`` != 0 on both ranks -- and the
    ghosts of the first rank were never overwritten with stale arena contents."""
    torch = gpu
    monkeypatch.setenv("FUS_IPC_SPIN_SECONDS", "1")
    scat, boxmesh, utils = pkg("scatterer"), pkg("boxmesh"), pkg("utils")
    meshes = [boxmesh.BoxMesh(2, (4, 2, 2), grid=(2, 1, 1), rank=r) for r in range(2)]
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    wid = next(_world_ids)
    comms = [scat.NativeComm(local=(wid, 2, r), transport="peer") for r in range(2)]
    fwd = [scat.scatter_forward(comms[r], od[r], gd[r], meshes[r].nlocal, np.float64) for r in range(2)]
    dev = torch.device("cuda", 0)
    b0 = torch.arange(meshes[0].ndofs, dtype=torch.float64, device=dev)
    b1 = torch.full((meshes[1].ndofs,), -5.0, dtype=torch.float64, device=dev)
    # a healthy exchange first (sequence 1)
    fwd[0].begin(b0), fwd[1].begin(b1)
    fwd[0].end(b0), fwd[1].end(b1)
    torch.cuda.synchronize()
    assert comms[0].health() == 0 and comms[1].health() == 0
    assert torch.equal(b1[meshes[1].nlocal:], b0[torch.from_numpy(np.asarray(gd[0][0])).to(dev)])
    b1[meshes[1].nlocal:] = -9.0  # consuming the arena's stale message 1 would overwrite these
    good = b1.clone()
    # sequence 2: rank 1 alone
    fwd[1].begin(b1)
    fwd[1].end(b1)
    torch.cuda.synchronize()
    st1 = fwd[1].status()
    assert st1["timeouts"] >= 1 and st1["dead"] and comms[1].health() >= 1
    assert torch.equal(b1, good)  # nothing was consumed from the arena
    # rank 0 posts its sequence 2 now: the credit it waits for is poisoned
    fwd[0].begin(b0)
    fwd[0].end(b0)
    torch.cuda.synchronize()
    st0 = fwd[0].status()
    assert st0["poisoned"] >= 1 and st0["timeouts"] == 0 and st0["dead"] and comms[0].health() >= 1
    for sc in fwd:
        sc.close()
    for c in comms:
        c.close()


def test_fork_join_enforce_one_caller_stream(gpu):
    """``fus_comm_fork`` / ``fus_comm_join`` have ONE sequence flag per direction: consecutive forks of a communicator must
    come from one caller stream.  Enforced: a join from another stream than its fork, and a fork
    from another stream while the communicator's stream still has forked work, are refused (FUS_ERR_INVALID_ARGUMENT,
    nothing launched); a change of the caller stream once the communicator's stream has drained is fine."""
    torch = gpu
    scat, lib_mod = pkg("scatterer"), pkg("_lib")
    comm = scat.NativeComm(transport="peer")
    side = comm.stream()
    s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.zeros(1 << 28, dtype=torch.float64, device="cuda")  # 2 GiB: a fill takes ~0.5 ms
    # the fill runs on the default stream and every stream below is non-blocking: without this the first adds overtake the fill on
    # part of the vector (what made this test fail now and then: tools/stream_visibility_probe.py, profiles/r05c_stream_visibility_probe.log)
    torch.cuda.synchronize()
    with torch.cuda.stream(s_a):
        comm.fork()
        with torch.cuda.stream(side):
            for _ in range(40):
                big.add_(1.0)  # ~40 ms of work on the communicator's stream, ordered after the fork
    with torch.cuda.stream(s_b):
        with pytest.raises(lib_mod.FusGpuError):
            comm.join()  # not the stream that forked
        with pytest.raises(lib_mod.FusGpuError):
            comm.fork()  # the communicator's stream is still busy with work forked from s_a
    with torch.cuda.stream(s_a):
        comm.join()
    torch.cuda.synchronize()
    assert float(big[0].item()) == 40.0
    with torch.cuda.stream(s_b):  # drained: another caller stream may take over
        comm.fork()
        with torch.cuda.stream(side):
            big.add_(1.0)
        comm.join()
    side.synchronize()
    torch.cuda.synchronize()
    wrong = int((big != 41.0).sum().item())  # the whole vector, counted on the device: says how much of the last add is missing
    assert wrong == 0 and comm.health() == 0, f"{wrong} of {big.numel()} elements != 41 (min {float(big.min())}, max {float(big.max())})"
    del big
    comm.close()
