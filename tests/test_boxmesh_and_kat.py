"""Synthetic mesh / partition plumbing and operator known-answer tests through the
oracle (the invariants a dolfinx assembly would also satisfy)."""

import numpy as np
import pytest

from conftest import build_problem, pkg, rel_l2
from oracle import oracle_np


@pytest.mark.parametrize("P,cells,grid", [(2, (4, 3, 2), (2, 1, 1)), (3, (4, 4, 3), (2, 2, 1)), (2, (2, 2, 2), (2, 2, 2)),
                                          (1, (5, 4, 3), (2, 2, 1)), (4, (3, 2, 2), (1, 2, 1))])
def test_partition_consistency(P, cells, grid):
    boxmesh, utils = pkg("boxmesh"), pkg("utils")
    R = int(np.prod(grid))
    meshes = [boxmesh.BoxMesh(P, cells, grid=grid, rank=r) for r in range(R)]
    nglob = int(np.prod([P * c + 1 for c in cells]))
    owned = np.concatenate([m.global_lexicographic_ids()[: m.nlocal] for m in meshes])
    assert owned.size == nglob and np.array_equal(np.sort(owned), np.arange(nglob))  # every dof owned exactly once
    assert sum(m.ncells for m in meshes) == int(np.prod(cells))
    lex_of_global = {}
    for m in meshes:
        im = m.index_map
        assert im.size_global == nglob and im.local_range[1] - im.local_range[0] == m.nlocal
        lex = m.global_lexicographic_ids()
        for li in range(m.nlocal):
            lex_of_global[im.local_range[0] + li] = lex[li]
    for m in meshes:
        im, lex = m.index_map, m.global_lexicographic_ids()
        # ghost global index and owner agree with the owner's numbering
        for g in range(m.nghost):
            assert lex_of_global[int(im.ghosts[g])] == lex[m.nlocal + g]
            o = int(im.owners[g])
            assert meshes[o].index_map.local_range[0] <= im.ghosts[g] < meshes[o].index_map.local_range[1]
            assert o < m.rank  # lowest rank touching a dof owns it
        # boundary cells first, and exactly they touch ghosts
        touches = (m.dofmap >= m.nlocal).any(axis=1)
        assert np.all(touches[: m.num_boundary_cells]) and not np.any(touches[m.num_boundary_cells:])
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    for r, m in enumerate(meshes):
        assert int(np.sum(od[r][1])) == m.nghost
        # dest ranks adjacency matches the ghosting ranks found by the exchange
        assert sorted(np.unique(m.index_map.index_to_dest_ranks().array).tolist()) == sorted(gd[r][3].tolist())


def test_partitioned_apply_equals_serial_in_process(oracle_c):
    """P-rank partitioned apply + reverse scatter == 1-rank apply (simulated ranks)."""
    boxmesh, utils = pkg("boxmesh"), pkg("utils")
    from halo_cpu import global_cell_constants

    P, cells, grid = 2, (4, 4, 2), (2, 2, 1)
    ser = build_problem(P, cells, perturb=0.16, seed=3)
    y_ser = np.zeros(ser["mesh"].ndofs)
    oracle_c.stiffness_apply(P, ser["D"], ser["x"], global_cell_constants(ser["mesh"]), y_ser, ser["G"], ser["mesh"].dofmap)
    R = int(np.prod(grid))
    parts = [build_problem(P, cells, perturb=0.16, seed=3, grid=grid, rank=r) for r in range(R)]
    meshes = [p["mesh"] for p in parts]
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    nl = [m.nlocal for m in meshes]
    xs = []
    for p in parts:
        x = p["x"].copy()
        x[p["mesh"].nlocal:] = np.nan  # stale ghosts
        xs.append(x)
    oracle_np.scatter_forward_all(xs, od, gd, nl)
    ys = []
    for p, x in zip(parts, xs):
        y = np.zeros(p["mesh"].ndofs)
        oracle_c.stiffness_apply(P, p["D"], x, global_cell_constants(p["mesh"]), y, p["G"], p["mesh"].dofmap)
        ys.append(y)
    oracle_np.scatter_reverse_all(ys, od, gd, nl)
    for m, y in zip(meshes, ys):
        lex = m.global_lexicographic_ids()[: m.nlocal]
        assert rel_l2(y[: m.nlocal], y_ser[lex]) < 1e-14


@pytest.mark.parametrize("P", [2, 4, 5])
def test_operator_known_answers(oracle_c, P):
    """mass(1) sums to the volume; K const = 0; 1^T K u = 0; u^T K u = |a|^2 vol for u = a.x;
    symmetry; positive semi-definiteness."""
    pb = build_problem(P, (3, 2, 2), perturb=0.0, random_constants=False)
    mesh = pb["mesh"]
    n = P + 1
    ones = np.ones(mesh.ndofs)
    m = np.zeros(mesh.ndofs)
    oracle_c.mass_apply(ones, pb["cc"], m, pb["detJ"], mesh.dofmap)
    assert abs(m.sum() - 1.0) < 1e-13

    def K(u):
        y = np.zeros(mesh.ndofs)
        oracle_c.stiffness_apply(P, pb["D"], u, pb["cc"], y, pb["G"], mesh.dofmap)
        return y

    assert np.max(np.abs(K(3.0 * ones))) < 1e-11
    u = pb["x"]
    Ku = K(u)
    assert abs(Ku.sum()) < 1e-9 * np.abs(Ku).sum()
    a = np.array([1.5, -2.0, 0.5])
    lin = mesh.dof_coordinates() @ a
    assert abs(lin @ K(lin) - a @ a * 1.0) < 1e-11
    v = np.random.default_rng(0).standard_normal(mesh.ndofs)
    assert abs(v @ Ku - u @ K(v)) < 1e-10 * abs(v @ Ku)
    assert v @ K(v) > 0


def test_energy_convergence(oracle_c):
    """u^T K u -> int |grad u|^2 = 100^2 (4+9+16) pi^2 / 8 for the reference's test field."""
    exact = 100.0**2 * 29 * np.pi**2 / 8
    errs = []
    for N in (2, 4):
        pb = build_problem(4, N, random_constants=False)
        y = np.zeros(pb["mesh"].ndofs)
        oracle_c.stiffness_apply(4, pb["D"], pb["x"], pb["cc"], y, pb["G"], pb["mesh"].dofmap, threads=2)
        errs.append(abs(pb["x"] @ y - exact) / exact)
    assert errs[1] < 1e-3 and errs[1] < errs[0] / 20


def test_facet_tables():
    boxmesh = pkg("boxmesh")
    m = boxmesh.BoxMesh(2, (2, 2, 2))
    lfd = m.local_facet_dofs()
    assert lfd.shape == (6, 9)
    assert sorted(lfd[0].tolist()) == [l for l in range(27) if l % 3 == 0]  # z = 0 face: k = 0
    assert sorted(lfd[3].tolist()) == list(range(18, 27))  # x = 1 face: i = 2
    bd = m.boundary_facets()
    assert bd.shape == (24, 2)
    fd = m.facet_dofmap(m.boundary_facets([2]))
    # x = 0 face dofs have lexicographic x-index 0 -> global ids below (2P+1)^2
    assert fd.max() < 25


def test_boundary_first_cell_order():
    boxmesh, utils = pkg("boxmesh"), pkg("utils")
    m = boxmesh.BoxMesh(2, (4, 4, 2), grid=(2, 2, 1), rank=3)
    rng = np.random.default_rng(0)
    shuffle = rng.permutation(m.ncells)
    dm = m.dofmap[shuffle]
    perm, nb = utils.boundary_first_cell_order(dm, m.nlocal)
    assert nb == m.num_boundary_cells
    d2 = dm[perm]
    touches = (d2 >= m.nlocal).any(axis=1)
    assert np.all(touches[:nb]) and not np.any(touches[nb:])
    assert sorted(perm.tolist()) == list(range(m.ncells))


@pytest.mark.parametrize("P,cells,cpb", [(4, (6, 5, 12), 10), (2, (5, 6, 7), 28), (3, (4, 6, 9), 16)])
def test_two_row_strip_order_is_a_permutation_with_more_sharing(P, cells, cpb):
    """plan_tiles (opt-in set-up order of the scatter-bound kernels' plans): face adjacency from the six face-interior dofs of the
    dofmap alone, rows of the cell order, adjacent rows interleaved.  The result is a permutation of the cells; consecutive cells of a
    strip are face neighbours or row neighbours; batches of ``cpb`` cells touch fewer distinct dofs than rows do."""
    boxmesh, pt = pkg("boxmesh"), pkg("plan_tiles")
    mesh = boxmesh.BoxMesh(P, cells)
    n = P + 1
    faces = pt.face_interior_local_dofs(n)
    assert faces.size == 6 and np.unique(faces).size == 6
    nbr = pt.face_neighbours(mesh.dofmap[:, faces])
    # a structured box: cell (i, j, k) has 6 neighbours minus the faces on the boundary; adjacency is symmetric through opposite faces
    nx, ny, nz = cells
    assert int((nbr >= 0).sum()) == 2 * ((nx - 1) * ny * nz + nx * (ny - 1) * nz + nx * ny * (nz - 1))
    c, f = np.nonzero(nbr >= 0)
    assert np.array_equal(nbr[nbr[c, f], f ^ 1], c)
    order = pt.two_row_strip_order(mesh.dofmap[:, faces])
    assert order is not None and np.array_equal(np.sort(order), np.arange(mesh.ncells))

    def distinct(o):
        return sum(np.unique(mesh.dofmap[o[b: b + cpb]]).size for b in range(0, mesh.ncells, cpb))

    # (rows shorter than a batch: a batch holds several whole rows either way -- equal; rows longer than a batch: strictly fewer)
    assert distinct(order) <= distinct(np.arange(mesh.ncells)) and (cells[2] < cpb or distinct(order) < distinct(np.arange(mesh.ncells)))
    # a cell order without rows (random) offers nothing to pair
    perm = np.random.default_rng(0).permutation(mesh.ncells)
    assert pt.two_row_strip_order(mesh.dofmap[perm][:, faces]) is None
