"""demo_linear_box caller (BASELINE config 3 shape, small size): the GPU RK4 solver -- both the
reference launch sequence and the fused stage -- against the oracle-side solver."""

import numpy as np
import pytest

from conftest import pkg, rel_l2
from oracle import rk4_oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused", [False, True], ids=["reference-sequence", "fused"])
@pytest.mark.parametrize("source_time", ["tn", "t"])
def test_linear_box_pressure_field(oracle_c, fused, source_time):
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls = pkg("boxmesh"), pkg("linear_solver")
    P, N, L = 4, 6, 0.012
    mesh = boxmesh.BoxMesh(P, N, length=L)
    h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    assert abs(h - np.sqrt(3) * L / N) < 1e-12
    dt, tf, nstep = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    nsteps = 12
    solver = ls.LinearSpectral3D(mesh, np.float64, fused=fused, source_time=source_time)
    solver.init()
    t, steps = solver.rk4(0.0, tf, dt, max_steps=nsteps)
    assert steps == nsteps and abs(t - nsteps * dt) < 1e-15
    u_ref, v_ref = rk4_oracle.solve(mesh, nsteps, dt, source_time=source_time, oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    assert rel_l2(solver.u_sol(), u_ref) < 1e-11
    assert rel_l2(solver.v_sol(), v_ref) < 1e-11


def test_fused_matches_reference_sequence_perturbed_mesh():
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls = pkg("boxmesh"), pkg("linear_solver")
    mesh = boxmesh.BoxMesh(3, (5, 4, 4), length=0.01, perturb=0.1, seed=2)
    h = ls.time_step_parameters(mesh, 3, 1500.0, 0.5e6, 0.01)
    dt, tf, _ = ls.snap_time_step(h, 3, 1500.0, 0.5e6, 0.01)
    res = []
    for fused, geom in ((False, False), (True, False), (True, True), (False, True)):
        s = ls.LinearSpectral3D(mesh, np.float64, fused=fused, in_kernel_geometry=geom)
        assert s.in_kernel_geometry == geom
        s.init()
        s.rk4(0.0, tf, dt, max_steps=20)
        res.append((s.u_sol(), s.v_sol()))
    for r in res[1:]:  # fused, fused + G formed in the kernel, reference sequence + G formed in the kernel
        assert rel_l2(r[0], res[0][0]) < 1e-12 and rel_l2(r[1], res[0][1]) < 1e-12


def _bowl_warp(xg):
    """Smooth non-affine map (SURVEY 8d geometry iii): the x = 0 face bulges like a bowl."""
    L = 0.006
    y, z = xg[:, 1] / L - 0.5, xg[:, 2] / L - 0.5
    out = xg.copy()
    out[:, 0] = xg[:, 0] + 0.15 * L * (y * y + z * z) * (1.0 - xg[:, 0] / L)
    return out


@pytest.mark.parametrize("fused", [False, True, "geom", "two-gather", "geom-two-gather"],
                         ids=["reference-sequence", "fused", "fused-in-kernel-geometry", "fused-two-gather", "fused-geom-two-gather"])
@pytest.mark.parametrize("P,cells", [(6, (3, 2, 2)), (4, (4, 3, 3)), (2, (7, 5, 6))], ids=["P6", "P4", "P2"])
def test_westervelt_bowl_pressure_field(oracle_c, P, cells, fused):
    """BASELINE config 5 shape (Westervelt, curved trilinear cells, P = 6) at test size; "geom": the fused
    cell pass forms G and detJ from the cell vertices instead of reading the precomputed arrays."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls, nls = pkg("boxmesh"), pkg("linear_solver"), pkg("nonlinear_solver")
    L = 0.006
    mesh = boxmesh.BoxMesh(P, cells, length=L, warp=_bowl_warp)
    h = ls.time_step_parameters(mesh, P, 1480.0, 1.1e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 1480.0, 1.1e6, L)
    nsteps = 10
    s = nls.WesterveltSpectral3D(mesh, np.float64, fused=bool(fused), in_kernel_geometry=str(fused).startswith("geom"),
                                 uniform_ratio=False if str(fused).endswith("two-gather") else True)
    assert (s.kappa is not None) == (fused in (True, "geom") or fused is False)  # uniform_ratio=True: single gather on this homogeneous medium
    s.init()
    s.rk4(0.0, tf, dt, max_steps=nsteps)
    u_ref, v_ref = rk4_oracle.solve_westervelt(mesh, nsteps, dt, oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    assert rel_l2(s.u_sol(), u_ref) < 1e-11
    assert rel_l2(s.v_sol(), v_ref) < 1e-11


def test_cuda_flavour_driver_calls_verbatim(oracle_c):
    """The reference's CUDA driver, call for call (cuda/demo_linear_box.py:364-575): ``cuda.to_device``,
    ``kernel[grid, block](...)`` launches with its launch configurations, 2-D ``dphi``, the
    3-element scatter data (single rank: empty), ``copy_to_host`` -- on the package's device shim and
    operators, against the oracle-side solver (source evaluated at ``t`` like the CUDA demos)."""
    boxmesh, gll, pre = pkg("boxmesh"), pkg("gll"), pkg("precompute")
    cuda = pkg("device")
    ops, sc, utils = pkg("operators"), pkg("scatterer"), pkg("utils")
    mass_operator, stiffness_operator = ops.mass_operator, ops.stiffness_operator
    axpy, copy, fill, pointwise_divide = ops.axpy, ops.copy, ops.fill, ops.pointwise_divide
    cuda.select_device(0)
    float_type = np.float64
    basis_degree, L = 3, 0.012
    nd = basis_degree + 1
    mesh = boxmesh.BoxMesh(basis_degree, (4, 3, 3), length=L)
    speed_of_sound, density, source_frequency, source_amplitude = 1500.0, 1000.0, 0.5e6, 60000.0
    angular_frequency = 2 * np.pi * source_frequency
    dofmap, num_cells = mesh.dofmap, mesh.ncells
    nlocal, ndofs = mesh.nlocal, mesh.ndofs
    owners_data, ghosts_data = utils.compute_scatterer_data(mesh.index_map)

    class _SelfComm:  # one rank: nothing to exchange
        rank, size = 0, 1

    scatter_rev = sc.scatter_reverse(_SelfComm(), owners_data, ghosts_data, nlocal, float_type)
    scatter_fwd = sc.scatter_forward(_SelfComm(), owners_data, ghosts_data, nlocal, float_type)
    pts, wts, dphi_1D = gll.tabulate_1d(basis_degree)
    w3, w2 = gll.tensor_weights_3d(wts), gll.tensor_weights_2d(wts)
    dphi = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))
    nq = w3.size
    detJ = np.zeros((num_cells, nq))
    G = np.zeros((num_cells, nq, 6))
    pre.compute_scaled_jacobian_determinant(detJ, (mesh.x_dofs, mesh.x_g), num_cells, dphi, w3)
    pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), num_cells, dphi, w3)
    bd1, bd2 = mesh.boundary_facets([2]), mesh.boundary_facets([3])
    dphi_f = pre.tabulate_facet_gradients(pts)
    detJ_f1, detJ_f2 = np.zeros((bd1.shape[0], nd * nd)), np.zeros((bd2.shape[0], nd * nd))
    pre.compute_boundary_facets_scaled_jacobian_determinant(detJ_f1, (mesh.x_dofs, mesh.x_g), bd1, dphi_f, w2)
    pre.compute_boundary_facets_scaled_jacobian_determinant(detJ_f2, (mesh.x_dofs, mesh.x_g), bd2, dphi_f, w2)
    bfacet_dofmap1, bfacet_dofmap2 = mesh.facet_dofmap(bd1), mesh.facet_dofmap(bd2)
    cell_coeff1 = np.full(num_cells, 1.0 / density / speed_of_sound**2)
    cell_coeff2 = np.full(num_cells, -1.0 / density)
    facet_coeff1 = np.full(bd1.shape[0], 1.0 / density)
    facet_coeff2 = np.full(bd2.shape[0], -1.0 / density / speed_of_sound)

    # ---- host to device (:364-385) ----
    cell_coeff1_d, cell_coeff2_d = cuda.to_device(cell_coeff1), cuda.to_device(cell_coeff2)
    dofmap_d, detJ_d, G_d, dphi_1D_d = cuda.to_device(dofmap), cuda.to_device(detJ), cuda.to_device(G), cuda.to_device(dphi_1D)
    facet_coeff1_d, facet_coeff2_d = cuda.to_device(facet_coeff1), cuda.to_device(facet_coeff2)
    bfacet_dofmap1_d, bfacet_dofmap2_d = cuda.to_device(bfacet_dofmap1), cuda.to_device(bfacet_dofmap2)
    detJ_f1_d, detJ_f2_d = cuda.to_device(detJ_f1), cuda.to_device(detJ_f2)
    zeros = np.zeros(ndofs)
    u_t_d, g_d, u_n_d, v_n_d, m_d, b_d = (cuda.to_device(zeros) for _ in range(6))
    # ---- launch configurations exactly as the demo computes them (:389-410); ignored by the library ----
    threadsperblock_m = 128
    num_blocks_m = (dofmap.size + (threadsperblock_m - 1)) // threadsperblock_m
    num_blocks_f1 = (bfacet_dofmap1.size + (threadsperblock_m - 1)) // threadsperblock_m
    num_blocks_f2 = (bfacet_dofmap2.size + (threadsperblock_m - 1)) // threadsperblock_m
    threadsperblock_s, num_blocks_s = (nd, nd, nd), num_cells
    threadsperblock_dofs = 1024
    num_blocks_dofs = (ndofs + (threadsperblock_dofs - 1)) // threadsperblock_dofs
    stiff_operator_cell = stiffness_operator(basis_degree, float_type)
    # ---- LHS (:421-428) ----
    fill[num_blocks_dofs, threadsperblock_dofs](1.0, u_t_d)
    fill[num_blocks_dofs, threadsperblock_dofs](0.0, m_d)
    mass_operator[num_blocks_m, threadsperblock_m](u_t_d, cell_coeff1_d, m_d, detJ_d, dofmap_d)
    cuda.synchronize()
    scatter_rev(m_d)
    # ---- RK4 (:437-566) ----
    a_runge, b_runge, c_runge = [0.0, 0.5, 0.5, 1.0], [1 / 6, 1 / 3, 1 / 3, 1 / 6], [0.0, 0.5, 0.5, 1.0]
    u_d, v_d, un_d, vn_d, u0_d, v0_d, ku_d, kv_d = (cuda.to_device(zeros) for _ in range(8))
    h = pkg("linear_solver").time_step_parameters(mesh, basis_degree, speed_of_sound, source_frequency, L)
    dt, _, _ = pkg("linear_solver").snap_time_step(h, basis_degree, speed_of_sound, source_frequency, L)
    t, nsteps = 0.0, 6
    for _ in range(nsteps):
        copy[num_blocks_dofs, threadsperblock_dofs](u_d, u0_d)
        copy[num_blocks_dofs, threadsperblock_dofs](v_d, v0_d)
        for i in range(4):
            copy[num_blocks_dofs, threadsperblock_dofs](u0_d, un_d)
            copy[num_blocks_dofs, threadsperblock_dofs](v0_d, vn_d)
            axpy[num_blocks_dofs, threadsperblock_dofs](a_runge[i] * dt, ku_d, un_d)
            axpy[num_blocks_dofs, threadsperblock_dofs](a_runge[i] * dt, kv_d, vn_d)
            copy[num_blocks_dofs, threadsperblock_dofs](vn_d, ku_d)
            T, alpha = 1 / source_frequency, 4.0
            window = 0.5 * (1.0 - np.cos(source_frequency * np.pi * t / alpha)) if t < T * alpha else 1.0
            g_vals = window * source_amplitude * angular_frequency / speed_of_sound * np.cos(angular_frequency * t)
            fill[num_blocks_dofs, threadsperblock_dofs](g_vals, g_d)
            copy[num_blocks_dofs, threadsperblock_dofs](un_d, u_n_d)
            copy[num_blocks_dofs, threadsperblock_dofs](vn_d, v_n_d)
            cuda.synchronize()
            scatter_fwd(u_n_d)
            scatter_fwd(v_n_d)
            fill[num_blocks_dofs, threadsperblock_dofs](0.0, b_d)
            stiff_operator_cell[num_blocks_s, threadsperblock_s](u_n_d, cell_coeff2_d, b_d, G_d, dofmap_d, dphi_1D_d)
            mass_operator[num_blocks_f1, threadsperblock_m](g_d, facet_coeff1_d, b_d, detJ_f1_d, bfacet_dofmap1_d)
            mass_operator[num_blocks_f2, threadsperblock_m](v_n_d, facet_coeff2_d, b_d, detJ_f2_d, bfacet_dofmap2_d)
            cuda.synchronize()
            scatter_rev(b_d)
            pointwise_divide[num_blocks_dofs, threadsperblock_dofs](b_d, m_d, kv_d)
            axpy[num_blocks_dofs, threadsperblock_dofs](b_runge[i] * dt, ku_d, u_d)
            axpy[num_blocks_dofs, threadsperblock_dofs](b_runge[i] * dt, kv_d, v_d)
        t += dt
    cuda.synchronize()
    u_host = u_d.copy_to_host()
    u_ref, _ = rk4_oracle.solve(mesh, nsteps, dt, source_time="t", oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    assert rel_l2(u_host, u_ref) < 1e-11


def test_linear_box_complete_run(oracle_c):
    """A complete demo_linear_box run (start to final time, ~66 RK4 steps, source ramp and the shorter last step included) on a
    small box: fused GPU solver vs the oracle-side solver, pressure field at the final time."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls = pkg("boxmesh"), pkg("linear_solver")
    P, N, L = 3, 5, 0.012
    mesh = boxmesh.BoxMesh(P, N, length=L)
    h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    dt, tf, nstep = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    solver = ls.LinearSpectral3D(mesh, np.float64, fused=True)
    solver.init()
    t, steps = solver.rk4(0.0, tf, dt)
    dts = rk4_oracle.step_sizes(0.0, tf, dt)  # same loop as the reference: the last step may be shorter
    assert steps == len(dts) and abs(t - tf) < 1e-12 and steps >= nstep - 1
    u_ref, v_ref = rk4_oracle.solve(mesh, steps, dts, oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 1e3  # the wave has crossed the box
    assert rel_l2(solver.u_sol(), u_ref) < 1e-10
    assert rel_l2(solver.v_sol(), v_ref) < 1e-10


def test_driver_scripts_run(tmp_path):
    """The demo_linear_box / time_operators counterparts stay runnable end to end."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    pkgdir = os.path.join(ROOT, "fenicsx-fus-gpu_amd")
    out = os.path.join(tmp_path, "plane.npz")
    ev = os.path.join(tmp_path, "eval")
    r = subprocess.run([sys.executable, os.path.join(pkgdir, "demo_linear_box.py"), "--cells", "6", "--degree", "3",
                        "--max-steps", "5", "--out", out, "--eval-out", ev], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "Solve time per step" in r.stdout, r.stdout + r.stderr
    d = np.load(out)
    assert d["u"].size == (3 * 6 + 1) ** 2 and int(d["steps"]) == 5
    # the reference's output: 100 x 100 points of the z = 0 plane, rows x,y,value (cuda/demo_linear_box.py:128-141,587-605);
    # points that coincide with dofs of the plane carry those dofs' values
    rows = np.loadtxt(os.path.join(ev, "pressure_field_nproc1.txt"), delimiter=",")
    assert rows.shape == (10000, 3) and abs(rows[:, 0].max() - 0.12) < 1e-8 and np.max(np.abs(rows[:, 2])) > 0
    corner = rows[(rows[:, 0] == 0) & (rows[:, 1] == 0)][0, 2]
    assert abs(corner - d["u"][np.argmin(d["lex"])]) < 1e-7
    r = subprocess.run([sys.executable, os.path.join(pkgdir, "time_operators.py"), "--degree", "2", "--cells", "6",
                        "--nreps", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.count("Elapsed time") == 3, r.stdout + r.stderr


def test_nonlinear_bowl_driver_dumps_last_period(tmp_path, oracle_c):
    """demo_nonlinear_bowl.py (counterpart of cuda/demo_nonlinear_bowl.py): complete run on a small bowl-warped box;
    once t > L/c + 6/f it writes one pressure-field file per time step for exactly one period (:662-680), in the
    reference's ``x,y,value`` text format; the last dumped field equals the oracle-side Westervelt loop run to the same
    step."""
    import os
    import subprocess
    import sys

    from conftest import ROOT

    out_dir = os.path.join(tmp_path, "fields")
    P, N, L = 3, 4, 0.004
    r = subprocess.run([sys.executable, os.path.join(ROOT, "fenicsx-fus-gpu_amd", "demo_nonlinear_bowl.py"), "--degree", str(P), "--cells", str(N),
                        "--length", str(L), "--out-dir", out_dir], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "Solve time per step" in r.stdout, r.stdout + r.stderr
    lines = {ln.split(":")[0]: ln.split(":")[1].strip() for ln in r.stdout.splitlines() if ln.startswith("Number of")}
    spp, nstep = int(lines["Number of steps per period"]), int(lines["Number of steps"])
    files = sorted(os.listdir(out_dir), key=lambda f: int(f.split("_")[-1][:-4]))
    assert files == [f"pressure_field_{k}.txt" for k in range(spp)]  # one period, no more
    last = np.loadtxt(os.path.join(out_dir, files[-1]), delimiter=",")
    assert last.shape == ((P * N + 1) ** 2, 3)
    # the oracle-side loop on the same mesh up to the step of the last dump
    boxmesh, ls = pkg("boxmesh"), pkg("linear_solver")

    def bowl(xg):
        out = xg.copy()
        yy, zz = xg[:, 1] / L - 0.5, xg[:, 2] / L - 0.5
        out[:, 0] = xg[:, 0] + 0.15 * (L / N) * 4 * (yy * yy + zz * zz) * (1.0 - xg[:, 0] / L)
        return out

    mesh = boxmesh.BoxMesh(P, N, length=L, warp=bowl)
    c0, f0 = 1480.0, 1.1e6
    h = ls.time_step_parameters(mesh, P, c0, f0, L)
    dt = 0.40 * h / (c0 * P**2)
    assert int((1 / f0) / dt) + 1 == spp
    dt = (1 / f0) / spp
    first = int(np.floor((L / c0 + 6.0 / f0) / dt)) + 1  # first step whose end time exceeds the collection threshold
    while not first * dt > L / c0 + 6.0 / f0:
        first += 1
    k_last = first + spp - 1
    assert k_last <= nstep
    u_ref, _ = rk4_oracle.solve_westervelt(mesh, k_last, dt, c0=c0, f0=f0, oracle_c=oracle_c)
    lex = mesh.global_lexicographic_ids()
    gd = mesh.global_dof_dims
    sel = np.nonzero((lex % gd[2]) == gd[2] // 2)[0]
    assert np.max(np.abs(u_ref[sel])) > 1.0
    # the file holds 8 decimals (the reference's "%.8f")
    assert np.max(np.abs(last[:, 2] - u_ref[sel])) < 1e-7 + 1e-9 * np.max(np.abs(u_ref[sel]))


def _lockstep(gens):
    """Advance the generators of several in-process ranks together: each ``next`` runs a rank up to the point
    where it has posted a set of halo exchanges; results are the generators' return values."""
    out = [None] * len(gens)
    live = list(enumerate(gens))
    while live:
        nxt = []
        for i, g in live:
            try:
                next(g)
                nxt.append((i, g))
            except StopIteration as done:
                out[i] = done.value
        live = nxt
    return out


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "reference-sequence"])
@pytest.mark.parametrize("grid,ghost_order", [((2, 1, 1), "owner"), ((2, 2, 1), 5)], ids=["2ranks", "4ranks-permuted-ghosts"])
def test_partitioned_linear_solver_async_transport_one_gpu(oracle_c, grid, ghost_order, fused):
    """The linear RK4 solver on 2 / 4 ranks sharing cuda:0 in this process with the exchange issued from
    C++ on its own stream (in-process transport, no host synchronisation): the set-up reverse scatter of the
    lumped mass, per stage the grouped forward scatter of (u_n, v_n) | interior cells | boundary cells +
    facet terms | reverse scatter of b | interior cells, stage kinds FIRST / LAST -- against the single-rank
    oracle-side solver."""
    import itertools

    import torch

    torch.cuda.set_device(0)
    boxmesh, ls, scat, utils = pkg("boxmesh"), pkg("linear_solver"), pkg("scatterer"), pkg("utils")
    P, cells, L = 3, (4, 4, 4), 0.012
    R = int(np.prod(grid))
    meshes = [boxmesh.BoxMesh(P, cells, grid=grid, rank=r, length=L, ghost_order=ghost_order) for r in range(R)]
    serial = boxmesh.BoxMesh(P, cells, length=L)
    h = ls.time_step_parameters(serial, P, 1500.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    wid = 7000 + 10 * R + int(fused)
    solvers = [ls.LinearSpectral3D(meshes[r], np.float64, comm=scat.NativeComm(local=(wid, R, r)), fused=fused,
                                   halo_plan=(od[r], gd[r]), defer_setup_exchange=True) for r in range(R)]
    _lockstep([s._setup for s in solvers])
    for s in solvers:
        s.init()
    res = _lockstep([s.rk4_schedule(0.0, tf, dt, max_steps=8) for s in solvers])
    torch.cuda.synchronize()
    assert all(r[1] == 8 for r in res)
    u_ref, _ = rk4_oracle.solve(serial, 8, dt, oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    seen = np.zeros(u_ref.size, dtype=int)
    for m, s in zip(meshes, solvers):
        lex = m.global_lexicographic_ids()[: m.nlocal]
        seen[lex] += 1
        assert rel_l2(s.u_sol(), u_ref[lex]) < 1e-11
    assert np.all(seen == 1)


def test_partitioned_reference_sequence_large_facet_sets_use_atomic_facet_mass(oracle_c):
    """in the reference launch sequence on a partitioned mesh the two facet mass applies are
    ``boundary_terms`` of ``HaloApply`` -- in the concurrent schedule they run on the communicator's stream while the interior
    stiffness launch adds into the same ``b`` with float atomics.  From 32 768 facet entries up ``mass_operator`` takes the
    atomic-free gather kernel (plain load + store per dof), which would lose those adds; the solver must hand HaloApply the
    float-atomic twin.  P = 4, 4 x 37 x 37 cells on 2 in-process ranks over the PEER transport: 1 369 facets = 34 225 entries
    per face, interior cells of rank 0 on the x = 0 face, of rank 1 on the x = L face; against the serial oracle loop."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls, scat, utils, ops = pkg("boxmesh"), pkg("linear_solver"), pkg("scatterer"), pkg("utils"), pkg("operators")
    P, cells, grid, L = 4, (4, 37, 37), (2, 1, 1), (0.004, 0.037, 0.037)
    R = 2
    meshes = [boxmesh.BoxMesh(P, cells, grid=grid, rank=r, length=L) for r in range(R)]
    serial = boxmesh.BoxMesh(P, cells, length=L)
    h = ls.time_step_parameters(serial, P, 1500.0, 0.5e6, L[0])
    dt, tf, _ = ls.snap_time_step(h, P, 1500.0, 0.5e6, L[0])
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    wid = 7050
    solvers = [ls.LinearSpectral3D(meshes[r], np.float64, comm=scat.NativeComm(local=(wid, R, r), transport="peer"), fused=False,
                                   halo_plan=(od[r], gd[r]), defer_setup_exchange=True) for r in range(R)]
    for s in solvers:
        assert s.halo.schedule_kind == "concurrent"
        assert s._mass_facet_stage is s.mass_facet.atomic  # what the stage launches next to the interior cells
    # the premise: left to itself the facet operator of this size IS the gather kernel
    big = solvers[0].fdm1  # rank 0 holds the source face x = 0
    assert big.shape[0] * big.shape[1] >= ops._MASS_PLAN_MIN_ENTRIES
    assert ops.mass_kernel_name(big, solvers[0].ndofs) == "fus::mass_gather_kernel"
    assert ops.mass_kernel_name(big, solvers[0].ndofs, atomic=True) != "fus::mass_gather_kernel"
    _lockstep([s._setup for s in solvers])
    for s in solvers:
        s.init()
    res = _lockstep([s.rk4_schedule(0.0, tf, dt, max_steps=4) for s in solvers])
    torch.cuda.synchronize()
    assert all(r[1] == 4 for r in res)
    u_ref, _ = rk4_oracle.solve(serial, 4, dt, oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    for m, s in zip(meshes, solvers):
        lex = m.global_lexicographic_ids()[: m.nlocal]
        assert rel_l2(s.u_sol(), u_ref[lex]) < 1e-11
        assert s.halo.health() == 0


@pytest.mark.parametrize("geom", [False, True], ids=["general-G", "in-kernel-geometry"])
def test_partitioned_westervelt_solver_async_transport_one_gpu(oracle_c, geom):
    """The fused Westervelt solver (BASELINE config 5 shape) on 2 in-process ranks over the asynchronous
    native transport: grouped set-up scatter of the three assembled diagonals (m0, w2, w5), per stage the
    grouped forward scatter of (w, v_n) and the reverse scatter of b."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls, nls, scat, utils = pkg("boxmesh"), pkg("linear_solver"), pkg("nonlinear_solver"), pkg("scatterer"), pkg("utils")
    P, cells, L, grid = 3, (4, 3, 3), 0.012, (2, 1, 1)
    R = 2
    meshes = [boxmesh.BoxMesh(P, cells, grid=grid, rank=r, length=L, ghost_order=3) for r in range(R)]
    serial = boxmesh.BoxMesh(P, cells, length=L)
    h = ls.time_step_parameters(serial, P, 1500.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    wid = 7100 + int(geom)
    solvers = [nls.WesterveltSpectral3D(meshes[r], np.float64, speed_of_sound=1500.0, source_frequency=0.5e6,
                                        comm=scat.NativeComm(local=(wid, R, r)), fused=True, in_kernel_geometry=geom,
                                        halo_plan=(od[r], gd[r]), defer_setup_exchange=True) for r in range(R)]
    _lockstep([s._setup for s in solvers])
    for s in solvers:
        s.init()
    res = _lockstep([s.rk4_schedule(0.0, tf, dt, max_steps=8) for s in solvers])
    torch.cuda.synchronize()
    assert all(r[1] == 8 for r in res)
    u_ref, _ = rk4_oracle.solve_westervelt(serial, 8, dt, c0=1500.0, f0=0.5e6, oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    for m, s in zip(meshes, solvers):
        lex = m.global_lexicographic_ids()[: m.nlocal]
        assert rel_l2(s.u_sol(), u_ref[lex]) < 1e-11


@pytest.mark.parametrize("geom", ["affine", "perturbed", "perturbed-in-kernel-geometry"])
def test_linear_solver_graph_replay_matches_rk4(geom):
    """rk4_graph: the full-size steps replayed from one captured hipGraph (source values read from device
    memory, fus_facet_terms_dev_*), the shorter last step through rk4 -- same kernels, same order, same data
    as rk4 (equal up to the summation order of the operator's atomics, like two rk4 runs); a second call (graph reused)
    continues correctly."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls = pkg("boxmesh"), pkg("linear_solver")
    P, N, L = 3, 5, 0.012
    mesh = boxmesh.BoxMesh(P, N, length=L, perturb=0.0 if geom == "affine" else 0.12, seed=4)
    h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    a = ls.LinearSpectral3D(mesh, np.float64, in_kernel_geometry=geom.endswith("geometry"))
    b = ls.LinearSpectral3D(mesh, np.float64, in_kernel_geometry=geom.endswith("geometry"))
    a.init()
    b.init()
    ta, sa = a.rk4(0.0, tf, dt)
    t1, s1 = b.rk4_graph(0.0, tf, dt, max_steps=10)  # captures
    t2, s2 = b.rk4_graph(t1, tf, dt)                 # replays the same graph, then the short last step
    assert s1 == 10 and s1 + s2 == sa and t2 == ta
    assert len(b._graphs) == 1
    assert np.max(np.abs(a.u_sol())) > 1e3
    # atomics make the operator's sums order-dependent from run to run: equal up to that, not to a tolerance
    # of the method (two rk4 runs differ by the same amount)
    assert rel_l2(b.u_sol(), a.u_sol()) < 1e-13 and rel_l2(b.v_sol(), a.v_sol()) < 1e-13


def test_graph_replay_survives_plan_cache_eviction():
    """A captured step holds raw pointers into batch-plan workspaces; the plan cache is bounded (16) and evicts
    oldest-first.  The graph keeps its workspaces alive itself: after the cache has been flushed and more than 16
    other plans have been built (whose allocations would otherwise land on the freed memory), replay still equals rk4."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls, ops = pkg("boxmesh"), pkg("linear_solver"), pkg("operators")
    P, N, L = 3, 5, 0.012
    mesh = boxmesh.BoxMesh(P, N, length=L, perturb=0.12, seed=4)
    h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    a, b = ls.LinearSpectral3D(mesh, np.float64), ls.LinearSpectral3D(mesh, np.float64)
    a.init(), b.init()
    ta, sa = a.rk4(0.0, tf, dt)
    t1, s1 = b.rk4_graph(0.0, tf, dt, max_steps=5)  # captures
    assert len(b._graph_plans[dt]) >= 1
    held = [ws.data_ptr() for ws, _ in b._graph_plans[dt]]
    ops._PLANS.clear()  # every cached plan released; only the graph's references keep its workspaces alive
    dev = torch.device("cuda", 0)
    junk = []
    for k in range(ops._PLANS.capacity + 4):  # fresh plans of the same size class: they would reuse freed blocks
        dm = torch.from_numpy(np.roll(mesh.dofmap, k + 1, axis=0).copy()).to(dev)
        junk.append(dm)
        ws, _ = ops._PLANS.get(dm)
        assert ws.data_ptr() not in held
        ws.fill_(0xFF) if k % 2 else None  # scribble over some of them
        if k % 2:
            ops._PLANS.clear()
    torch.cuda.synchronize()
    t2, s2 = b.rk4_graph(t1, tf, dt)
    assert s1 + s2 == sa and t2 == ta
    assert rel_l2(b.u_sol(), a.u_sol()) < 1e-13 and rel_l2(b.v_sol(), a.v_sol()) < 1e-13


@pytest.mark.parametrize("mode", ["single-gather", "two-gather", "in-kernel-geometry"])
def test_westervelt_solver_graph_replay_matches_rk4(mode):
    """rk4_graph of the Westervelt solver (g and dg/dt of every stage read from device memory)."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, nls = pkg("boxmesh"), pkg("nonlinear_solver")
    P, cells, L = 4, (4, 3, 3), 0.006
    mesh = boxmesh.BoxMesh(P, cells, length=L, warp=_bowl_warp)
    kw = dict(fused=True, in_kernel_geometry=(mode == "in-kernel-geometry"), uniform_ratio=(False if mode == "two-gather" else True))
    a = nls.WesterveltSpectral3D(mesh, np.float64, **kw)
    b = nls.WesterveltSpectral3D(mesh, np.float64, **kw)
    assert (a.kappa is None) == (mode == "two-gather")
    dt = 0.4 * (L / 4) / (a.c0 * P * P)
    tf = 30.5 * dt  # 30 full steps and a shorter last one
    a.init()
    b.init()
    ta, sa = a.rk4(0.0, tf, dt)
    t1, s1 = b.rk4_graph(0.0, tf, dt, max_steps=7)
    t2, s2 = b.rk4_graph(t1, tf, dt)
    assert sa == 31 and s1 == 7 and s1 + s2 == sa and abs(t2 - ta) < 1e-18 and len(b._graphs) == 1
    assert np.max(np.abs(a.u_sol())) > 0
    assert rel_l2(b.u_sol(), a.u_sol()) < 1e-13 and rel_l2(b.v_sol(), a.v_sol()) < 1e-13


def test_config3_full_size_rk4_fused_vs_reference_sequence():
    """BASELINE config 3 at FULL size (demo_linear_box: P = 4, 54^3 perturbed cells, 10 218 313 dofs, general G): 5 RK4 steps
    of the fused path (one vector kernel per stage, facet terms in one launch, solution kept in (u0, v0)) against the
    reference's own launch sequence (cuda/demo_linear_box.py:487-566: 12 launches per stage) through the reference-compatible
    operators -- rel l2 <= 1e-12 -- plus the in-kernel-geometry variant, plus one invariant: with the source switched off the
    field stays exactly zero (the discrete operator maps 0 to 0: no stray contribution from plan slots or facets)."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls = pkg("boxmesh"), pkg("linear_solver")
    P, N, L = 4, 54, 0.12
    mesh = boxmesh.BoxMesh(P, N, length=L, perturb=0.16, seed=0)
    assert mesh.ndofs == 10218313
    h = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 1500.0, 0.5e6, L)
    res = {}
    for name, kw in (("reference", dict(fused=False)), ("fused", dict(fused=True)), ("fused-geom", dict(fused=True, in_kernel_geometry=True))):
        s = ls.LinearSpectral3D(mesh, np.float64, **kw)
        assert not s.affine
        s.init()
        t, steps = s.rk4(0.0, tf, dt, max_steps=5)
        assert steps == 5
        res[name] = (s.u_sol(), s.v_sol())
        del s
        torch.cuda.empty_cache()
    assert np.max(np.abs(res["reference"][0])) > 0
    for name in ("fused", "fused-geom"):
        assert rel_l2(res[name][0], res["reference"][0]) < 1e-12, name
        assert rel_l2(res[name][1], res["reference"][1]) < 1e-12, name
    s = ls.LinearSpectral3D(mesh, np.float64, fused=True, source_amplitude=0.0)
    s.init()
    s.rk4(0.0, tf, dt, max_steps=2)
    assert float(s.u.abs().max().item()) == 0.0 and float(s.v.abs().max().item()) == 0.0


def test_config3_and_config5_default_time_loops_vs_oracle_at_full_size(oracle_c):
    """the solvers' DEFAULT paths (fused stage, lean vector pass, G formed in the cell kernel) at the sizes their numbers are
    quoted on, DIRECTLY against the oracle's time loops (oracle/rk4_oracle.py over oracle/fus_oracle.c, all host cores; pinned by
    tests/golden/rk4*.npz) -- config 3: P = 4, 54^3 perturbed cells, 3 steps; config 5's shape: Westervelt, P = 6, 36^3 bowl-warped cells, 2 steps.
    The oracle reads the reference's G array, formed by a second solver instance that keeps it (keep_G=True)."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls, nls = pkg("boxmesh"), pkg("linear_solver"), pkg("nonlinear_solver")
    threads = max(1, min(32, oracle_c.max_threads()))
    h_ = lambda t: np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float64))  # noqa: E731
    # ---- config 3, linear
    P, N, L = 4, 54, 0.12
    mesh = boxmesh.BoxMesh(P, N, length=L, perturb=0.16, seed=0)
    hm = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(hm, P, 1500.0, 0.5e6, L)
    s = ls.LinearSpectral3D(mesh, np.float64)
    assert s.fused and s.in_kernel_geometry and s.lean_stages and s.G_array is None  # the default path
    s.init()
    _, steps = s.rk4(0.0, tf, dt, max_steps=3)
    assert steps == 3
    u_gpu, v_gpu = s.u_sol(), s.v_sol()
    del s
    g = ls.LinearSpectral3D(mesh, np.float64, in_kernel_geometry=False)
    geo = (h_(g.G_array), h_(g.detJ), h_(g.detJ_f1), h_(g.detJ_f2))
    del g
    torch.cuda.empty_cache()
    u_ref, v_ref = rk4_oracle.solve(mesh, 3, dt, oracle_c=oracle_c, threads=threads, geometry=geo)
    assert np.max(np.abs(u_ref)) > 0
    assert rel_l2(u_gpu, u_ref[: mesh.nlocal]) < 1e-11 and rel_l2(v_gpu, v_ref[: mesh.nlocal]) < 1e-11
    del geo, u_ref, v_ref
    # ---- config 5's shape, Westervelt
    P, N = 6, 36

    def bowl(xg):
        out = xg.copy()
        yy, zz = xg[:, 1] / L - 0.5, xg[:, 2] / L - 0.5
        out[:, 0] = xg[:, 0] + 0.15 * (L / N) * 4 * (yy * yy + zz * zz) * (1.0 - xg[:, 0] / L)
        return out

    mesh = boxmesh.BoxMesh(P, N, length=(L, L, L), warp=bowl)
    hm = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(hm, P, 1500.0, 0.5e6, L)
    s = nls.WesterveltSpectral3D(mesh, np.float64, speed_of_sound=1500.0, source_frequency=0.5e6, fused=True)
    assert s.in_kernel_geometry and s.lean_stages and s.kappa is None and s.G is None  # default: two-gather pass, G formed in the kernel
    s.init()
    _, steps = s.rk4(0.0, tf, dt, max_steps=2)
    assert steps == 2
    u_gpu, v_gpu = s.u_sol(), s.v_sol()
    del s
    g = nls.WesterveltSpectral3D(mesh, np.float64, speed_of_sound=1500.0, source_frequency=0.5e6, fused=True, in_kernel_geometry=False)
    geo = (h_(g.G), h_(g.detJ), h_(g.dF1), h_(g.dF2))
    del g
    torch.cuda.empty_cache()
    u_ref, v_ref = rk4_oracle.solve_westervelt(mesh, 2, dt, c0=1500.0, f0=0.5e6, oracle_c=oracle_c, threads=threads, geometry=geo)
    assert np.max(np.abs(u_ref)) > 0
    assert rel_l2(u_gpu, u_ref[: mesh.nlocal]) < 1e-11 and rel_l2(v_gpu, v_ref[: mesh.nlocal]) < 1e-11


def test_config5_full_degree_westervelt_fused_vs_reference_sequence():
    """BASELINE config 5's operator shape at a size one GPU holds comfortably (demo_nonlinear_bowl: Westervelt, P = 6, bowl-warped
    trilinear cells; 36^3 cells = 10 077 696 dofs): 3 RK4 steps of the fused path -- general G and G formed in the kernel --
    against the reference's launch sequence (four cell kernels per stage, cuda/demo_nonlinear_bowl.py:603-632)."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls, nls = pkg("boxmesh"), pkg("linear_solver"), pkg("nonlinear_solver")
    P, N, L = 6, 36, 0.06

    def bowl(xg):
        out = xg.copy()
        yy, zz = xg[:, 1] / L - 0.5, xg[:, 2] / L - 0.5
        out[:, 0] = xg[:, 0] + 0.15 * (L / N) * 4 * (yy * yy + zz * zz) * (1.0 - xg[:, 0] / L)
        return out

    mesh = boxmesh.BoxMesh(P, N, length=L, warp=bowl)
    assert mesh.ndofs == (P * N + 1) ** 3
    h = ls.time_step_parameters(mesh, P, 1480.0, 1.1e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 1480.0, 1.1e6, L)
    res = {}
    for name, kw in (("reference", dict(fused=False)), ("fused", dict(fused=True)), ("fused-geom", dict(fused=True, in_kernel_geometry=True))):
        s = nls.WesterveltSpectral3D(mesh, np.float64, **kw)
        s.init()
        _, steps = s.rk4(0.0, tf, dt, max_steps=3)
        assert steps == 3
        res[name] = (s.u_sol(), s.v_sol())
        del s
        torch.cuda.empty_cache()
    assert np.max(np.abs(res["reference"][0])) > 0
    for name in ("fused", "fused-geom"):
        assert rel_l2(res[name][0], res["reference"][0]) < 1e-11, name
        assert rel_l2(res[name][1], res["reference"][1]) < 1e-11, name


def _two_materials(mesh, L):
    """Per-cell material arrays of a two-material box: a slab of 'bone' (faster, denser, more absorbing, more nonlinear) across
    the middle third in x, 'water' around it (the DG0 arrays of the reference's production drivers, cuda/demo_nonlinear_bowl.py:166-178)."""
    xc = mesh.x_g[mesh.x_dofs].mean(axis=1)[:, 0]
    bone = (xc > L / 3) & (xc < 2 * L / 3)
    assert 0 < bone.sum() < mesh.ncells
    return dict(c=np.where(bone, 2800.0, 1480.0), rho=np.where(bone, 1850.0, 1000.0), beta=np.where(bone, 5.0, 3.5), att=np.where(bone, 2.0, 0.2))


@pytest.mark.parametrize("fused", [False, True], ids=["reference-sequence", "fused"])
def test_linear_solver_heterogeneous_medium(oracle_c, fused):
    """Per-cell speed of sound and density (scalars everywhere before round 4): the GPU solver, both stage implementations,
    against the oracle-side loop with the same arrays; the source term uses the coupling medium's scalar speed."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls = pkg("boxmesh"), pkg("linear_solver")
    P, cells, L = 3, (6, 3, 3), 0.012
    mesh = boxmesh.BoxMesh(P, cells, length=L, perturb=0.1, seed=2)
    mat = _two_materials(mesh, L)
    h = ls.time_step_parameters(mesh, P, 2800.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 2800.0, 0.5e6, L)
    s = ls.LinearSpectral3D(mesh, np.float64, speed_of_sound=mat["c"], density=mat["rho"], fused=fused)
    assert abs(s.c0 - 1480.0) < 1e-12  # the source facets lie in the water
    s.init()
    s.rk4(0.0, tf, dt, max_steps=12)
    u_ref, v_ref = rk4_oracle.solve(mesh, 12, dt, c0=mat["c"], rho0=mat["rho"], oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    assert rel_l2(s.u_sol(), u_ref) < 1e-11 and rel_l2(s.v_sol(), v_ref) < 1e-11
    with pytest.raises(ValueError):
        ls.LinearSpectral3D(mesh, np.float64, speed_of_sound=mat["c"][:-1])


@pytest.mark.parametrize("variant", ["reference-sequence", "fused", "fused-in-kernel-geometry"])
def test_westervelt_solver_heterogeneous_medium(oracle_c, variant):
    """Per-cell c, rho, beta and attenuation: c4 / c3 = delta / c^2 is not uniform any more, so the fused stage takes the
    two-gather cell pass by itself; against the oracle-side Westervelt loop with the same arrays."""
    import torch

    torch.cuda.set_device(0)
    boxmesh, ls, nls = pkg("boxmesh"), pkg("linear_solver"), pkg("nonlinear_solver")
    P, cells, L = 4, (6, 2, 2), 0.006
    mesh = boxmesh.BoxMesh(P, cells, length=L, warp=_bowl_warp)
    mat = _two_materials(mesh, L)
    h = ls.time_step_parameters(mesh, P, 2800.0, 1.1e6, L)
    dt, tf, _ = ls.snap_time_step(h, P, 2800.0, 1.1e6, L)
    s = nls.WesterveltSpectral3D(mesh, np.float64, speed_of_sound=mat["c"], density=mat["rho"], nonlinear_coefficient=mat["beta"],
                                 attenuation_coefficient_dB=mat["att"], fused=variant != "reference-sequence",
                                 in_kernel_geometry=variant.endswith("geometry"))
    assert s.kappa is None and abs(s.c0 - 1480.0) < 1e-12 and abs(s.rho0 - 1000.0) < 1e-12
    s.init()
    s.rk4(0.0, tf, dt, max_steps=10)
    u_ref, v_ref = rk4_oracle.solve_westervelt(mesh, 10, dt, c0=mat["c"], rho0=mat["rho"], beta=mat["beta"], att_dB=mat["att"], oracle_c=oracle_c)
    assert np.max(np.abs(u_ref)) > 0
    assert rel_l2(s.u_sol(), u_ref) < 1e-11 and rel_l2(s.v_sol(), v_ref) < 1e-11
