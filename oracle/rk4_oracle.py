"""ORACLE-side RK4 drivers (test infrastructure -- NOT product code): the linear wave solver of
numba-cpu/demo_linear_box.py:302-455 (f0 / f1 / RK4 loop, source evaluated at tn) and the Westervelt
solver of cuda/demo_nonlinear_bowl.py:357-374,458-475,540-650 on one rank, restated with the oracle's
operators.  Pinned against the reference-driven fixtures tests/golden/rk4_*.npz / rk4nl_*.npz
(tests/test_rk4_golden.py).  Used to check the GPU solvers' pressure fields -- in tests/ and, since round 6, as the CHECKER of every RK4 /
Westervelt step line of bench.py (benchlib/cpu_legs.py oracle_step_field) -- and by bench.py's ``cpu_baseline`` leg of the RK4-step line
(the reference prints "Solve time per step" of exactly this loop, numba-cpu/demo_linear_box.py:472-473).

Only tests/, __graft_entry__.smoke() and bench.py's checker / cpu_baseline legs (benchlib/) import this: never the product.  The mesh / table builders
it takes from the package (gll, precompute: host-side numpy, the counterparts of what the reference takes from
basix / dolfinx) are inputs, not the operators under test."""

import numpy as np

import fusgpu_loader
from oracle import oracle_np


def pkg(name):
    return fusgpu_loader.submodule(name)


A = (0.0, 0.5, 0.5, 1.0)
B = (1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0)
C = (0.0, 0.5, 0.5, 1.0)


def step_sizes(t0, tf, dt):
    """The reference's time loop (``while t < tf: dt = min(dt, tf - t); ...; t += dt``,
    cuda/demo_linear_box.py:487-488,566): the last step may be shorter."""
    out, t = [], float(t0)
    while t < tf:
        dt = min(dt, tf - t)
        out.append(dt)
        t += dt
    return out


def solve(mesh, nsteps, dt, c0=1500.0, rho0=1000.0, f0=0.5e6, p0=60000.0, source_time="tn", oracle_c=None, threads=1, timing=None,
          geometry=None, c_ref=None):
    """``dt`` may be a sequence of per-step sizes (then ``nsteps`` is ignored).  ``oracle_c``: the C restatement of the
    operators instead of the numpy one (``threads`` > 1: its OpenMP stiffness apply).  ``timing``: a dict that receives
    ``seconds_per_step`` (the time loop alone, set-up excluded -- what the reference prints as "Solve time per step").
    ``geometry = (G, detJ, detJ_f1, detJ_f2)``: geometry factors computed elsewhere (bench.py hands over the ones the GPU
    stepped with) instead of the host precompute."""
    import time
    gll, pre = pkg("gll"), pkg("precompute")
    P, n = mesh.P, mesh.P + 1
    pts, wts, D = gll.tabulate_1d(P)
    w3 = gll.tensor_weights_3d(wts)
    dg = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))
    nc = mesh.ncells
    bd1, bd2 = mesh.boundary_facets([2]), mesh.boundary_facets([3])
    if geometry is not None:
        G, detJ, dF1, dF2 = geometry
    else:
        G, detJ = np.zeros((nc, n**3, 6)), np.zeros((nc, n**3))
        pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), nc, dg, w3)
        pre.compute_scaled_jacobian_determinant(detJ, (mesh.x_dofs, mesh.x_g), nc, dg, w3)
        w2, dpf = gll.tensor_weights_2d(wts), pre.tabulate_facet_gradients(pts)
        dF1, dF2 = np.zeros((bd1.shape[0], n * n)), np.zeros((bd2.shape[0], n * n))
        pre.compute_boundary_facets_scaled_jacobian_determinant(dF1, (mesh.x_dofs, mesh.x_g), bd1, dpf, w2)
        pre.compute_boundary_facets_scaled_jacobian_determinant(dF2, (mesh.x_dofs, mesh.x_g), bd2, dpf, w2)
    fd1, fd2 = mesh.facet_dofmap(bd1), mesh.facet_dofmap(bd2)
    # c0 / rho0: scalars or one value per cell (heterogeneous medium); the source term uses the scalar ``c_ref``
    c_ref = float(c0) if np.ndim(c0) == 0 else (float(np.asarray(c0)[bd1[:, 0]].mean()) if c_ref is None else float(c_ref))
    c_, rho_ = (np.broadcast_to(np.asarray(a, dtype=np.float64), (nc,)).copy() for a in (c0, rho0))
    cc1, cc2 = 1 / rho_ / c_ / c_, -1 / rho_
    fc1, fc2 = 1 / rho_[bd1[:, 0]], -1 / rho_[bd2[:, 0]] / c_[bd2[:, 0]]
    c0 = c_ref
    nd = mesh.ndofs
    w0 = 2 * np.pi * f0
    m = np.zeros(nd)
    mass = oracle_c.mass_apply if oracle_c is not None else oracle_np.mass_apply
    mass(np.ones(nd), cc1, m, detJ, mesh.dofmap)

    def stiff(x, y):
        if oracle_c is not None:
            oracle_c.stiffness_apply(P, D, x, cc2, y, G, mesh.dofmap, threads=threads)
        else:
            oracle_np.stiffness_apply(P, D.flatten(), x, cc2, y, G, mesh.dofmap)

    def f1(t, un, vn):
        T, alpha = 1 / f0, 4.0
        window = 0.5 * (1 - np.cos(f0 * np.pi * t / alpha)) if t < T * alpha else 1.0
        g = np.full(nd, window * p0 * w0 / c0 * np.cos(w0 * t))
        b = np.zeros(nd)
        stiff(un, b)
        mass(g, fc1, b, dF1, fd1)
        mass(np.ascontiguousarray(vn), fc2, b, dF2, fd2)
        return b / m

    u, v = np.zeros(nd), np.zeros(nd)
    ku, kv = np.zeros(nd), np.zeros(nd)
    t = 0.0
    dts = list(dt) if hasattr(dt, "__len__") else [dt] * nsteps
    t_loop = time.perf_counter()
    for dt in dts:
        u0, v0 = u.copy(), v.copy()
        for i in range(4):
            un = u0 + A[i] * dt * ku
            vn = v0 + A[i] * dt * kv
            tn = t + C[i] * dt
            ku = vn.copy()
            kv = f1(tn if source_time == "tn" else t, un, vn)
            u = u + B[i] * dt * ku
            v = v + B[i] * dt * kv
        t += dt
    if timing is not None:
        timing["seconds_per_step"] = (time.perf_counter() - t_loop) / max(len(dts), 1)
        timing["steps"] = len(dts)
    return u, v


def solve_westervelt(mesh, nsteps, dt, c0=1480.0, rho0=1000.0, f0=1.1e6, p0=None, beta=3.5, att_dB=0.2,
                     source_time="tn", oracle_c=None, c_ref=None, rho_ref=None, threads=1, geometry=None):
    """cuda/demo_nonlinear_bowl.py:357-374,458-475,540-650 restated with the oracle's operators
    (single rank; source on x = 0, absorbing on x = L).  ``oracle_c``: the C restatement of the operators (``threads`` > 1: its
    OpenMP stiffness apply).  ``geometry = (G, detJ, detJ_f1, detJ_f2)``: geometry factors computed elsewhere (bench.py hands
    over the ones the GPU stepped with) instead of the host precompute."""
    gll, pre = pkg("gll"), pkg("precompute")
    P, n = mesh.P, mesh.P + 1
    # c0, rho0, beta, att_dB: scalars or one value per cell; the source term and the default amplitude use scalars (c_ref, rho_ref)
    bd_src = mesh.boundary_facets([2])
    c_ref = float(c0) if np.ndim(c0) == 0 else (float(np.asarray(c0)[bd_src[:, 0]].mean()) if c_ref is None else float(c_ref))
    rho_ref = float(rho0) if np.ndim(rho0) == 0 else (float(np.asarray(rho0)[bd_src[:, 0]].mean()) if rho_ref is None else float(rho_ref))
    if p0 is None:
        p0 = rho_ref * c_ref * 0.38557513826589934
    w0 = 2 * np.pi * f0
    c0, rho0, beta, att_dB = (np.broadcast_to(np.asarray(a, dtype=np.float64), (mesh.ncells,)).copy() for a in (c0, rho0, beta, att_dB))
    delta = 2 * (att_dB / 20 * np.log(10)) * c0**3 / w0 / w0
    pts, wts, D = gll.tabulate_1d(P)
    w3 = gll.tensor_weights_3d(wts)
    dg_ = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))
    nc = mesh.ncells
    bd1, bd2 = mesh.boundary_facets([2]), mesh.boundary_facets([3])
    if geometry is not None:
        G, detJ, dF1, dF2 = geometry
    else:
        G, detJ = np.zeros((nc, n**3, 6)), np.zeros((nc, n**3))
        pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), nc, dg_, w3)
        pre.compute_scaled_jacobian_determinant(detJ, (mesh.x_dofs, mesh.x_g), nc, dg_, w3)
        w2, dpf = gll.tensor_weights_2d(wts), pre.tabulate_facet_gradients(pts)
        dF1, dF2 = np.zeros((bd1.shape[0], n * n)), np.zeros((bd2.shape[0], n * n))
        pre.compute_boundary_facets_scaled_jacobian_determinant(dF1, (mesh.x_dofs, mesh.x_g), bd1, dpf, w2)
        pre.compute_boundary_facets_scaled_jacobian_determinant(dF2, (mesh.x_dofs, mesh.x_g), bd2, dpf, w2)
    fd1, fd2 = mesh.facet_dofmap(bd1), mesh.facet_dofmap(bd2)
    cc1 = 1 / rho0 / c0**2
    cc2 = -2 * beta / rho0**2 / c0**4
    cc3 = -1 / rho0
    cc4 = -delta / rho0 / c0**2
    cc5 = 2 * beta / rho0**2 / c0**4
    i1, i2 = bd1[:, 0], bd2[:, 0]
    f11, f21 = 1 / rho0[i1], delta[i1] / rho0[i1] / c0[i1] ** 2
    f12, f22 = delta[i2] / rho0[i2] / c0[i2] ** 3, -1 / rho0[i2] / c0[i2]
    c0 = c_ref  # from here on: the scalar of the source term
    nd = mesh.ndofs
    ones = np.ones(nd)
    m0 = np.zeros(nd)
    # the mass applies: the numpy restatement, or (large meshes) the C one -- both pinned by tests/test_oracle_golden.py
    mass_apply = oracle_c.mass_apply if (oracle_c is not None and geometry is not None) else oracle_np.mass_apply
    oracle_np.mass_apply(ones, cc1, m0, detJ, mesh.dofmap)
    oracle_np.mass_apply(ones, f12, m0, dF2, fd2)

    def stiff(x, cc, y):
        if oracle_c is not None:
            oracle_c.stiffness_apply(P, D, x, cc, y, G, mesh.dofmap, threads=threads)
        else:
            oracle_np.stiffness_apply(P, D.flatten(), x, cc, y, G, mesh.dofmap)

    def f1(t, un, vn):
        T, alpha = 1 / f0, 4.0
        if t < T * alpha:
            window = 0.5 * (1 - np.cos(f0 * np.pi * t / alpha))
            dwindow = 0.5 * np.pi * f0 / alpha * np.sin(f0 * np.pi * t / alpha)
        else:
            window, dwindow = 1.0, 0.0
        a = 2 * p0 * w0 / c0
        g = np.full(nd, window * a * np.cos(w0 * t))
        dg = np.full(nd, dwindow * a * np.cos(w0 * t) - window * a * w0 * np.sin(w0 * t))
        m = np.zeros(nd)
        mass_apply(np.ascontiguousarray(un), cc2, m, detJ, mesh.dofmap)
        m += m0
        b = np.zeros(nd)
        stiff(np.ascontiguousarray(un), cc3, b)
        stiff(np.ascontiguousarray(vn), cc4, b)
        mass_apply(vn * vn, cc5, b, detJ, mesh.dofmap)
        mass_apply(g, f11, b, dF1, fd1)
        mass_apply(dg, f21, b, dF1, fd1)
        mass_apply(np.ascontiguousarray(vn), f22, b, dF2, fd2)
        return b / m

    u, v = np.zeros(nd), np.zeros(nd)
    ku, kv = np.zeros(nd), np.zeros(nd)
    t = 0.0
    for _ in range(nsteps):
        u0, v0 = u.copy(), v.copy()
        for i in range(4):
            un = u0 + A[i] * dt * ku
            vn = v0 + A[i] * dt * kv
            tn = t + C[i] * dt
            ku = vn.copy()
            kv = f1(tn if source_time == "tn" else t, un, vn)
            u = u + B[i] * dt * ku
            v = v + B[i] * dt * kv
        t += dt
    return u, v
