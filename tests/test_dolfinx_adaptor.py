"""dolfinx adaptor on mock inputs shaped like what a dolfinx driver holds: the
function-space dofmap in the STANDARD basix local order with a ``dof_ordering`` permutation
(cuda/demo_linear_box.py:167-176), cells in arbitrary order, ghosts in arbitrary (not owner-grouped)
order.  dolfinx itself exists nowhere in this pipeline; BoxMesh supplies the ground truth the mock is
scrambled from.  CPU: adaptor invariants + partitioned apply == serial apply with the oracle as the
operator; GPU: the same through HaloApply, the HIP kernels and the native halo exchange."""
import itertools

import numpy as np
import pytest

from conftest import build_problem, pkg, rel_l2, ref_field
from halo_cpu import global_cell_constants

_world_ids = itertools.count(5000)


def mock_dolfinx_rank(P, cells, grid, rank, seed):
    """One rank's data as a dolfinx driver would see it, scrambled from BoxMesh."""
    boxmesh, gll, pre = pkg("boxmesh"), pkg("gll"), pkg("precompute")
    mesh = boxmesh.BoxMesh(P, cells, grid=grid, rank=rank, perturb=0.16, seed=3, ghost_order=seed)
    rng = np.random.default_rng(100 * seed + rank)
    n = P + 1
    cperm = rng.permutation(mesh.ncells)                     # cells in arbitrary order
    sigma = np.random.default_rng(seed).permutation(n**3)    # the element's dof_ordering (same on all ranks)
    dofmap_std = mesh.dofmap[cperm][:, sigma]                # V.dofmap.list on the standard element
    pts, wts, D = gll.tabulate_1d(P)
    G = np.zeros((mesh.ncells, n**3, 6))
    pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), mesh.ncells, pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts)),
                                          gll.tensor_weights_3d(wts))
    return dict(mesh=mesh, dofmap_std=dofmap_std, dof_ordering=sigma, G=G[cperm], cc=global_cell_constants(mesh)[cperm],
                x_dofs=mesh.x_dofs[cperm], D=D)


def test_tensor_product_dofmap_round_trip():
    ad = pkg("dolfinx_adaptor")
    r = mock_dolfinx_rank(3, (2, 2, 2), (1, 1, 1), 0, 7)
    tp = ad.tensor_product_dofmap(r["dofmap_std"], r["dof_ordering"])
    assert tp.dtype == np.int32 and tp.flags.c_contiguous
    # same cells (as sets of rows) as the ground-truth tensor-product dofmap
    assert sorted(map(tuple, tp)) == sorted(map(tuple, r["mesh"].dofmap))
    assert np.array_equal(ad.tensor_product_dofmap(r["mesh"].dofmap, None), r["mesh"].dofmap)
    with pytest.raises(ValueError):
        ad.tensor_product_dofmap(r["dofmap_std"], np.zeros(64, dtype=int))


@pytest.mark.parametrize("P,cells,grid", [(2, (4, 4, 2), (2, 2, 1)), (3, (4, 2, 2), (2, 1, 1)), (2, (4, 4, 4), (2, 2, 2))])
def test_partitioned_apply_through_adaptor_cpu(oracle_c, P, cells, grid):
    from oracle import oracle_np

    ad, utils = pkg("dolfinx_adaptor"), pkg("utils")
    R = int(np.prod(grid))
    ranks = [mock_dolfinx_rank(P, cells, grid, r, 11) for r in range(R)]
    rms = []
    for rk in ranks:
        tp = ad.tensor_product_dofmap(rk["dofmap_std"], rk["dof_ordering"])
        rm, (G, cc, xd) = ad.partition_for_overlap(tp, rk["mesh"].index_map, (rk["G"], rk["cc"], rk["x_dofs"]))
        nb = rm.num_boundary_cells
        assert (rm.dofmap[:nb] >= rm.nlocal).any(axis=1).all() and not (rm.dofmap[nb:] >= rm.nlocal).any()
        assert rm.ncells == rk["mesh"].ncells and rm.ndofs == rk["mesh"].ndofs
        rms.append((rm, G, cc))
    od, gd = utils.compute_scatterer_data_all([rm.index_map for rm, _, _ in rms])
    nl = [rm.nlocal for rm, _, _ in rms]
    xs = []
    for rk, (rm, _, _) in zip(ranks, rms):
        x = ref_field(rk["mesh"].dof_coordinates())
        x[rm.nlocal:] = -777.0
        xs.append(x)
    oracle_np.scatter_forward_all(xs, od, gd, nl)
    ys = []
    for rk, (rm, G, cc), x in zip(ranks, rms, xs):
        y = np.zeros(rm.ndofs)
        oracle_c.stiffness_apply(P, rk["D"], x, cc, y, G, rm.dofmap)
        ys.append(y)
    oracle_np.scatter_reverse_all(ys, od, gd, nl)
    pb = build_problem(P, cells, perturb=0.16, seed=3)
    ms = pb["mesh"]
    y_ser = np.zeros(ms.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], global_cell_constants(ms), y_ser, pb["G"], ms.dofmap)
    for rk, y in zip(ranks, ys):
        m = rk["mesh"]
        lex = m.global_lexicographic_ids()
        assert rel_l2(y[: m.nlocal], y_ser[lex[: m.nlocal]]) < 1e-13


@pytest.mark.gpu
@pytest.mark.parametrize("P,cells,grid", [(4, (4, 4, 2), (2, 2, 1)), (2, (4, 4, 4), (2, 2, 2))])
def test_partitioned_apply_through_adaptor_gpu(oracle_c, P, cells, grid):
    """The same through HaloApply: HIP stiffness kernel on the three cell sub-ranges (each with its own
    locality-ordered plan: the mock's cells are in random order), native halo exchange (in-process
    transport), ghosts not owner-grouped."""
    import torch

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ad, utils, scat, ops = pkg("dolfinx_adaptor"), pkg("utils"), pkg("scatterer"), pkg("operators")
    R = int(np.prod(grid))
    wid = next(_world_ids)
    ranks = [mock_dolfinx_rank(P, cells, grid, r, 11) for r in range(R)]
    parts = []
    for rk in ranks:
        tp = ad.tensor_product_dofmap(rk["dofmap_std"], rk["dof_ordering"])
        parts.append(ad.partition_for_overlap(tp, rk["mesh"].index_map, (rk["G"], rk["cc"])))
    od, gd = utils.compute_scatterer_data_all([rm.index_map for rm, _ in parts])
    op = ops.stiffness_operator(P, ranks[0]["D"].flatten(), np.float64)
    state = []
    for r, (rk, (rm, (G, cc))) in enumerate(zip(ranks, parts)):
        x = ref_field(rk["mesh"].dof_coordinates())
        x[rm.nlocal:] = -777.0
        halo = scat.HaloApply(rm, op, scat.NativeComm(local=(wid, R, r)), np.float64, plan=(od[r], gd[r]))
        state.append(dict(halo=halo, x=torch.from_numpy(x).to(dev), y=torch.zeros(rm.ndofs, dtype=torch.float64, device=dev),
                          cc=torch.from_numpy(cc).to(dev), G=torch.from_numpy(G).to(dev), dm=torch.from_numpy(rm.dofmap).to(dev)))
    live = [s["halo"].apply_schedule(s["x"], s["cc"], s["y"], s["G"], s["dm"]) for s in state]
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
    pb = build_problem(P, cells, perturb=0.16, seed=3)
    ms = pb["mesh"]
    y_ser = np.zeros(ms.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], global_cell_constants(ms), y_ser, pb["G"], ms.dofmap)
    for rk, s in zip(ranks, state):
        m = rk["mesh"]
        lex = m.global_lexicographic_ids()
        assert rel_l2(s["y"].cpu().numpy()[: m.nlocal], y_ser[lex[: m.nlocal]]) < 1e-12


class _AdjacencyList:
    """The members of dolfinx.graph.AdjacencyList a driver touches."""

    def __init__(self, array, offsets):
        self.array, self.offsets = np.asarray(array, dtype=np.int32), np.asarray(offsets, dtype=np.int32)

    def links(self, i):
        return self.array[self.offsets[i]: self.offsets[i + 1]]

    @property
    def num_nodes(self):
        return self.offsets.size - 1


class DolfinxIndexMapLike:
    """Nothing but what ``dolfinx.common.IndexMap`` exposes and the reference's compute_scatterer_data reads
    (cuda/utils.py:20-47): size_local, num_ghosts, ghosts (GLOBAL indices), owners, local_range,
    index_to_dest_ranks() -> AdjacencyList.  Built from plain arrays only."""

    def __init__(self, size_local, local_range, ghosts, owners, dest_array, dest_offsets):
        self.size_local, self.num_ghosts = int(size_local), int(len(ghosts))
        self.local_range = (int(local_range[0]), int(local_range[1]))
        self.ghosts, self.owners = np.asarray(ghosts, dtype=np.int64), np.asarray(owners, dtype=np.int32)
        self._dest = _AdjacencyList(dest_array, dest_offsets)

    def index_to_dest_ranks(self):
        return self._dest


def _as_dolfinx_like(im):
    dest = im.index_to_dest_ranks()
    return DolfinxIndexMapLike(im.size_local, im.local_range, np.array(im.ghosts), np.array(im.owners), np.array(dest.array),
                               np.array(dest.offsets))


@pytest.mark.parametrize("fixture", ["halo_plan_P2_4x4x2_grid2x2x1_7", "halo_plan_P2_4x4x4_grid2x2x2_lex", "halo_plan_P3_3x3x3_grid3x1x1_5"])
def test_duck_typed_index_map_end_to_end_on_reference_plans(oracle_c, fixture):
    """A dolfinx-like IndexMap made of plain arrays -> adaptor -> halo plan == what the REFERENCE's compute_scatterer_data
    returned for the same partition (tests/golden/halo_plan_*.npz) -> partitioned apply == serial apply."""
    import os

    from conftest import GOLDEN
    from oracle import oracle_np

    ad, utils, boxmesh, gll, pre = (pkg(m) for m in ("dolfinx_adaptor", "utils", "boxmesh", "gll", "precompute"))
    d = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    P, shape, grid = int(d["P"]), tuple(int(v) for v in d["shape"]), tuple(int(v) for v in d["grid"])
    go = str(d["ghost_order"])
    go = go if go in ("owner", "lex") else int(go)
    R = int(np.prod(grid))
    meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r, ghost_order=go, perturb=0.1, seed=2) for r in range(R)]
    ims = [_as_dolfinx_like(m.index_map) for m in meshes]
    # the reference-sort plan, element for element, from the duck-typed maps
    od, gd = utils.compute_scatterer_data_all(ims, stable=False)
    for r in range(R):
        assert np.array_equal(od[r][0], d[f"owners_idx_{r}"]) and np.array_equal(od[r][3], d[f"unique_owners_{r}"])
        assert np.array_equal(gd[r][0], d[f"ghosts_idx_{r}"]) and np.array_equal(gd[r][3], d[f"unique_ghosts_{r}"])
    # end to end: scrambled cells -> partition_for_overlap(duck-typed map) -> exchange + operator == serial
    n = P + 1
    pts, wts, D = gll.tabulate_1d(P)
    od, gd = utils.compute_scatterer_data_all(ims)
    xs, rms = [], []
    for r, m in enumerate(meshes):
        G = np.zeros((m.ncells, n**3, 6))
        pre.compute_scaled_geometrical_factor(G, (m.x_dofs, m.x_g), m.ncells, pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts)),
                                              gll.tensor_weights_3d(wts))
        cperm = np.random.default_rng(r).permutation(m.ncells)
        rm, (Gp, ccp) = ad.partition_for_overlap(m.dofmap[cperm], ims[r], (G[cperm], global_cell_constants(m)[cperm]))
        assert rm.index_map is ims[r] and rm.nlocal == m.nlocal and rm.nghost == m.nghost
        x = ref_field(m.dof_coordinates())
        x[m.nlocal:] = -777.0
        xs.append(x)
        rms.append((rm, Gp, ccp))
    nl = [m.nlocal for m in meshes]
    oracle_np.scatter_forward_all(xs, od, gd, nl)
    ys = []
    for (rm, Gp, ccp), x in zip(rms, xs):
        y = np.zeros(rm.ndofs)
        oracle_c.stiffness_apply(P, D, x, ccp, y, Gp, rm.dofmap)
        ys.append(y)
    oracle_np.scatter_reverse_all(ys, od, gd, nl)
    pb = build_problem(P, shape, perturb=0.1, seed=2)
    ms = pb["mesh"]
    y_ser = np.zeros(ms.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], global_cell_constants(ms), y_ser, pb["G"], ms.dofmap)
    for m, y in zip(meshes, ys):
        lex = m.global_lexicographic_ids()
        assert rel_l2(y[: m.nlocal], y_ser[lex[: m.nlocal]]) < 1e-13


# ---------------------------------------------------------------------------------------------- ArrayMesh
def scrambled_array_mesh(P, cells, grid=(1, 1, 1), rank=0, seed=0, L=0.012, perturb=0.12, warp=None, ghost_order="owner"):
    """A BoxMesh's rank handed over the way a dolfinx driver would: cells in random order, vertices randomly renumbered,
    the two tagged facet sets as ``(cell, local facet)`` pairs in THAT cell numbering (tags 1 = source, 2 = absorbing, like
    the reference's meshes) -- plain arrays only.  Returns (ArrayMesh, BoxMesh)."""
    boxmesh, ad = pkg("boxmesh"), pkg("dolfinx_adaptor")
    box = boxmesh.BoxMesh(P, cells, grid=grid, rank=rank, length=L, perturb=perturb, seed=5, warp=warp, ghost_order=ghost_order)
    rng = np.random.default_rng(1000 * seed + rank)
    cperm = rng.permutation(box.ncells)  # new cell c is box cell cperm[c]
    inv = np.empty(box.ncells, dtype=np.int64)
    inv[cperm] = np.arange(box.ncells)
    vperm = rng.permutation(box.x_g.shape[0])  # new vertex v is box vertex vperm[v]
    vinv = np.empty_like(vperm)
    vinv[vperm] = np.arange(vperm.size)
    tags = {}
    for tag, face in ((1, 2), (2, 3)):
        bd = box.boundary_facets([face])
        tags[tag] = np.stack([inv[bd[:, 0]], bd[:, 1]], axis=1) if bd.shape[0] else np.zeros((0, 2), np.int32)
    am = ad.ArrayMesh(P, box.dofmap[cperm], vinv[box.x_dofs[cperm]], box.x_g[vperm], index_map=box.index_map if np.prod(grid) > 1 else None,
                      facet_tags=tags, ndofs_global=box.ndofs_global)
    return am, box


@pytest.mark.parametrize("grid,rank", [((1, 1, 1), 0), ((2, 1, 1), 0), ((2, 1, 1), 1), ((2, 2, 1), 3)])
def test_array_mesh_invariants(grid, rank):
    """ArrayMesh from scrambled plain arrays: same cells, geometry and facet dofmaps as the structured mesh it came from,
    ghost-touching cells first, facet cells remapped with the cells."""
    am, box = scrambled_array_mesh(3, (4, 4, 2), grid=grid, rank=rank, seed=3)
    assert am.ncells == box.ncells and am.nlocal == box.nlocal and am.ndofs == box.ndofs
    assert sorted(map(tuple, am.dofmap)) == sorted(map(tuple, box.dofmap))
    nb = am.num_boundary_cells
    assert nb == box.num_boundary_cells
    assert (am.dofmap[:nb] >= am.nlocal).any(axis=1).all() and not (am.dofmap[nb:] >= am.nlocal).any()
    # geometry follows the cells: the vertex coordinates of every cell are those of the box cell with the same dofs
    key = {tuple(row): c for c, row in enumerate(box.dofmap)}
    for c in range(am.ncells):
        cb = key[tuple(am.dofmap[c])]
        assert np.array_equal(am.x_g[am.x_dofs[c]], box.x_g[box.x_dofs[cb]])
    for tag, face in ((1, 2), (2, 3)):
        fa, fb = am.facet_dofmap(am.boundary_facets([tag])), box.facet_dofmap(box.boundary_facets([face]))
        assert fa.shape == fb.shape and sorted(map(tuple, fa)) == sorted(map(tuple, fb))
    assert am.boundary_facets([99]).shape == (0, 2)
    ad = pkg("dolfinx_adaptor")
    with pytest.raises(ValueError):
        ad.ArrayMesh(3, am.dofmap[:, :10], am.x_dofs, am.x_g)
    with pytest.raises(ValueError):
        ad.ArrayMesh(3, am.dofmap, am.x_dofs, am.x_g, facet_tags={1: np.array([[am.ncells, 0]])})


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["linear-fused", "linear-reference", "linear-in-kernel-geometry", "westervelt-fused", "westervelt-geom"])
def test_solvers_step_an_array_mesh_one_rank(solver):
    """the solver classes on a mesh handed over as plain arrays (randomly renumbered cells and vertices,
    tagged facets as (cell, local facet) pairs) reproduce the structured-mesh run: owned field equal to 1e-12 (the dof
    numbering is the same, so the fields compare entry by entry)."""
    import torch

    torch.cuda.set_device(0)
    ls, nls = pkg("linear_solver"), pkg("nonlinear_solver")
    P, cells, L = (3, (5, 4, 3), 0.012)
    warp = None
    if solver.startswith("westervelt"):
        def warp(xg):  # noqa: E306  (the bowl of BASELINE config 5 at test size)
            out = xg.copy()
            y, z = xg[:, 1] / L - 0.5, xg[:, 2] / L - 0.5
            out[:, 0] = xg[:, 0] + 0.15 * L * (y * y + z * z) * (1.0 - xg[:, 0] / L)
            return out
    am, box = scrambled_array_mesh(P, cells, seed=2, L=L, warp=warp)
    c0, f0 = (1480.0, 1.1e6) if solver.startswith("westervelt") else (1500.0, 0.5e6)
    h = ls.time_step_parameters(box, P, c0, f0, L)
    assert abs(ls.time_step_parameters(am, P, c0, f0, L) - h) < 1e-15
    dt, tf, _ = ls.snap_time_step(h, P, c0, f0, L)
    out = []
    for mesh in (box, am):
        if solver.startswith("linear"):
            s = ls.LinearSpectral3D(mesh, np.float64, fused=solver != "linear-reference", in_kernel_geometry=solver.endswith("geometry"))
        else:
            s = nls.WesterveltSpectral3D(mesh, np.float64, fused=True, in_kernel_geometry=solver.endswith("geom"))
        s.init()
        _, steps = s.rk4(0.0, tf, dt, max_steps=10)
        assert steps == 10
        out.append((s.u_sol(), s.v_sol()))
    assert np.max(np.abs(out[0][0])) > 0
    assert rel_l2(out[1][0], out[0][0]) < 1e-12 and rel_l2(out[1][1], out[0][1]) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["local", "peer"])
def test_linear_solver_steps_array_meshes_two_in_process_ranks(transport):
    """Two in-process ranks, each an ArrayMesh (scrambled cells / vertices, ghosts NOT numbered owner by owner, duck-typed
    index map): the partitioned fused solver over the asynchronous transports == the one-rank structured solver."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls, scat, utils = pkg("boxmesh"), pkg("linear_solver"), pkg("scatterer"), pkg("utils")
    P, cells, L, grid = 3, (4, 4, 3), 0.012, (2, 1, 1)
    pairs = [scrambled_array_mesh(P, cells, grid=grid, rank=r, seed=4, L=L, ghost_order=9) for r in range(2)]
    meshes = [am for am, _ in pairs]
    serial = boxmesh.BoxMesh(P, cells, length=L, perturb=0.12, seed=5)
    h = ls.time_step_parameters(serial, P, 1500.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    wid = next(_world_ids)
    comms = [scat.NativeComm(local=(wid, 2, r), transport="peer" if transport == "peer" else "rccl") for r in range(2)]
    solvers = [ls.LinearSpectral3D(meshes[r], np.float64, comm=comms[r], fused=True, halo_plan=(od[r], gd[r]), defer_setup_exchange=True)
               for r in range(2)]

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

    lockstep([s._setup for s in solvers])
    for s in solvers:
        s.init()
    lockstep([s.rk4_schedule(0.0, tf, dt, max_steps=8) for s in solvers])
    torch.cuda.synchronize()
    for s in solvers:
        s.check_halo_health()
    ref = ls.LinearSpectral3D(serial, np.float64, fused=True)
    ref.init()
    ref.rk4(0.0, tf, dt, max_steps=8)
    u_ref = ref.u_sol()
    assert np.max(np.abs(u_ref)) > 0
    for (am, box), s in zip(pairs, solvers):
        lex = box.global_lexicographic_ids()[: box.nlocal]
        assert rel_l2(s.u_sol(), u_ref[lex]) < 1e-11


@pytest.mark.gpu
def test_array_mesh_heterogeneous_materials_follow_the_cell_order():
    """Per-cell material arrays are given in the CALLER's cell order; ``ArrayMesh`` re-orders its cells and the solver permutes
    the arrays with them: the scrambled mesh with the scrambled arrays reproduces the structured run."""
    import torch

    torch.cuda.set_device(0)
    ls = pkg("linear_solver")
    P, cells, L = 3, (6, 3, 3), 0.012
    am, box = scrambled_array_mesh(P, cells, seed=6, L=L)
    xc = box.x_g[box.x_dofs].mean(axis=1)[:, 0]
    bone = (xc > L / 3) & (xc < 2 * L / 3)
    c_box, rho_box = np.where(bone, 2800.0, 1480.0), np.where(bone, 1850.0, 1000.0)
    # the caller's (scrambled) cell order: identify each of the caller's cells by its dofs
    key = {tuple(row): i for i, row in enumerate(box.dofmap)}
    caller_dofmap = np.empty_like(am.dofmap)
    caller_dofmap[am.cell_permutation] = am.dofmap  # ArrayMesh cell c is the caller's cell cell_permutation[c]
    to_box = np.array([key[tuple(r)] for r in caller_dofmap])
    h = ls.time_step_parameters(box, P, 2800.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 2800.0, 0.5e6, L)
    out = []
    for mesh, c, rho in ((box, c_box, rho_box), (am, c_box[to_box], rho_box[to_box])):
        s = ls.LinearSpectral3D(mesh, np.float64, speed_of_sound=c, density=rho, fused=True)
        s.init()
        s.rk4(0.0, tf, dt, max_steps=10)
        out.append(s.u_sol())
    assert np.max(np.abs(out[0])) > 0 and rel_l2(out[1], out[0]) < 1e-12
