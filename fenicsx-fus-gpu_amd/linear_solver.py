"""
Linear acoustic wave solver (explicit RK4 on u' = v, M v' = -K u + boundary terms) on the
synthetic box mesh, driven entirely by this package's operators and scatterers -- the
"demo_linear_box" caller of the hot path (BASELINE config 3).

Semantics follow the reference drivers:
  set-up        cuda/demo_linear_box.py:245-345 (geometry factors, source facets x = 0,
                absorbing facets x = L, material coefficients), :421-428 (lumped mass
                m = M(1/(rho c^2)) 1, scatter_rev)
  time step     cuda/demo_linear_box.py:115-122 (dt = CFL h / (c P^2), snapped to an integer
                number of steps per period; final time L/c + 2/f)
  RK4 stage     cuda/demo_linear_box.py:487-566; source window / g evaluated at the STAGE
                time tn as fenicsx/demo_linear_box.py:187-199,270, numba-cpu/demo_linear_box.py:345-358
                and cpp/common/Linear.hpp:179-187,317 do (the CUDA demos use t -- SURVEY 3.4 quirks;
                ``source_time="t"`` reproduces them)
The class shape mirrors cpp/common/Linear.hpp:52-348 (LinearSpectral3D: init / rk4 / u_sol).

Two stage implementations with identical results up to round-off:
  fused=False   the reference's exact launch sequence (5 copy, 4 axpy, 2 fill, 1 divide, 3 operator
                applies, 3 scatters per stage), through the reference-compatible call surface;
  fused=True    one fused vector kernel per stage (csrc/rk4.hpp), ku doubling as v_n, the source
                term through scaled facet constants instead of a full-vector fill, 1/m precomputed,
                halo exchange overlapped with interior cells.
"""

from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from . import operators as ops
from .gll import gll_points_weights, tabulate_1d, tensor_points_3d, tensor_weights_2d, tensor_weights_3d
from .step_graph import StepGraphMixin
from .precompute import (
    compute_boundary_facets_scaled_jacobian_determinant_device,
    compute_scaled_geometrical_factor_device,
    tabulate_facet_gradients,
    tabulate_hex_p1_gradients,
)


def per_cell(value, mesh, name):
    """A material parameter as a per-cell array in the MESH's cell order: a scalar (homogeneous medium, the reference's box
    demos) or one value per cell in the caller's cell order (the DG0 arrays ``c0.x.array`` ... of the reference's production
    drivers, cuda/demo_nonlinear_bowl.py:166-178; a mesh that re-ordered its cells -- ``ArrayMesh`` -- permutes them)."""
    a = np.asarray(value, dtype=np.float64)
    if a.ndim == 0:
        return np.full(mesh.ncells, float(a))
    if a.shape != (mesh.ncells,):
        raise ValueError(f"{name}: a scalar or one value per cell ({mesh.ncells}), got shape {a.shape}")
    return np.ascontiguousarray(mesh.permute_cells(a) if hasattr(mesh, "permute_cells") else a)


def device_geometry(mesh, P, ft, dev, facet_sets):
    """G, detJ and the facet detJ of the given boundary_data sets, computed on the device
    (csrc/geometry.hpp; the reference does this with numba on the host,
    cuda/demo_linear_box.py:245-317)."""
    import torch

    from .gll import tabulate_1d, tensor_points_3d, tensor_weights_2d, tensor_weights_3d

    n = P + 1
    pts, wts, D = tabulate_1d(P, ft)
    td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    tdt = torch.float64 if np.dtype(ft) == np.float64 else torch.float32
    w3, w2 = td(tensor_weights_3d(wts).astype(ft)), td(tensor_weights_2d(wts).astype(ft))
    dphi_g = td(tabulate_hex_p1_gradients(tensor_points_3d(pts), ft))
    dphi_f = td(tabulate_facet_gradients(pts, ft))
    gm = (td(mesh.x_dofs), td(mesh.x_g))
    G = torch.empty((mesh.ncells, n**3, 6), dtype=tdt, device=dev)
    detJ = torch.empty((mesh.ncells, n**3), dtype=tdt, device=dev)
    compute_scaled_geometrical_factor_device(G, gm, mesh.ncells, dphi_g, w3, detJ=detJ)
    out = []
    for bd in facet_sets:
        dF = torch.zeros((bd.shape[0], n * n), dtype=tdt, device=dev)
        if bd.shape[0]:
            compute_boundary_facets_scaled_jacobian_determinant_device(dF, gm, td(bd.astype(np.int32)), dphi_f, w2)
        out.append(dF)
    return D, G, detJ, out

A_RUNGE = (0.0, 0.5, 0.5, 1.0)
B_RUNGE = (1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0)
C_RUNGE = (0.0, 0.5, 0.5, 1.0)


def time_step_parameters(mesh, P, speed_of_sound, source_frequency, domain_length, CFL=0.65):
    """cuda/demo_linear_box.py:115-122.  ``mesh_size`` = smallest cell diameter (dolfinx
    ``cpp.mesh.h``: largest vertex-vertex distance of a cell), min over ranks by the caller."""
    xg = mesh.x_g.astype(np.float64)[mesh.x_dofs]  # [nc, 8, 3]
    d = np.linalg.norm(xg[:, :, None, :] - xg[:, None, :, :], axis=-1).reshape(mesh.ncells, -1).max(axis=1)
    return float(d.min())


def snap_time_step(mesh_size, P, speed_of_sound, source_frequency, domain_length, CFL=0.65):
    period = 1.0 / source_frequency
    dt = CFL * mesh_size / (speed_of_sound * P**2)
    step_per_period = int(period / dt) + 1
    dt = period / step_per_period
    final_time = domain_length / speed_of_sound + 2.0 / source_frequency
    return dt, final_time, int((final_time - 0.0) / dt) + 1


class LinearSpectral3D(StepGraphMixin):
    def __init__(self, mesh, float_type=np.float64, speed_of_sound=1500.0, density=1000.0,
                 source_frequency=0.5e6, source_amplitude=60000.0, comm=None, fused=True,
                 source_time="tn", overlap=True, halo_kernels=None, affine="auto", in_kernel_geometry="auto",
                 halo_plan=None, defer_setup_exchange=False, reference_speed_of_sound=None, keep_G=False):
        """``speed_of_sound`` / ``density``: scalars, or one value per cell (heterogeneous medium: the DG0 material arrays of
        the reference's drivers, in the caller's cell order).  ``reference_speed_of_sound``: the c of the source term
        ``p0 w0 / c cos(w0 t)`` (cuda/demo_linear_box.py:515-530 uses the scalar of its homogeneous medium); default: the
        scalar given, or the mean over the cells of the source facets.
        ``in_kernel_geometry``: ``"auto"`` (default) -- on non-affine cells of degree >= 3 the stiffness apply forms G in the kernel
        from the 8 vertices of each cell (the reference's geometry is P1 everywhere, cuda/demo_nonlinear_bowl.py:317): -18 % per
        step and no 6 n^3-value-per-cell G array at config 3; ``False`` keeps the reference's G stream
        (numba-cpu/precompute.py:115-163), ``True`` forces the kernel form for any degree.  ``keep_G``: keep the G array on the
        device although the apply does not read it (``solver.G_array``)."""
        self.mesh, self.P = mesh, mesh.P
        self.dt_np = np.dtype(float_type)
        self.tdt = _lib.torch_dtype(float_type)
        self.tdt_np = self.dt_np
        c_cells, rho_cells = per_cell(speed_of_sound, mesh, "speed_of_sound"), per_cell(density, mesh, "density")
        self.c0, self.rho0 = float(c_cells.mean()), float(rho_cells.mean())
        self.f0, self.p0 = float(source_frequency), float(source_amplitude)
        self.w0 = 2.0 * np.pi * self.f0
        self.fused, self.source_time = bool(fused), source_time
        self.lean_stages = os.environ.get("FUS_RK4_LEAN", "1") != "0"  # the fused stage's vector pass: kinds 4-7 of csrc/rk4.hpp (_stage_args)
        if comm is not None:  # an MPI.Comm (the reference's comm = MPI.COMM_WORLD) becomes the bootstrap of a NativeComm
            from .scatterer import as_comm

            comm = as_comm(comm)
        self.comm = comm
        P, n = self.P, self.P + 1
        dev = torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        ft = self.dt_np

        # ---- geometry precompute (reference: numba on the host, cuda/demo_linear_box.py:245-317) --
        nc = mesh.ncells
        # the two tagged facet sets (cuda/demo_linear_box.py:230-243, cuda/utils.py:81-114): a structured box names them by its
        # faces (x = 0: source, x = L: absorbing), a mesh handed over as arrays (dolfinx_adaptor.ArrayMesh) by its facet tags
        bd1 = mesh.boundary_facets([getattr(mesh, "source_tag", 2)])
        bd2 = mesh.boundary_facets([getattr(mesh, "absorbing_tag", 3)])
        D, G_d, detJ_d, (dF1_d, dF2_d) = device_geometry(mesh, P, ft, dev, (bd1, bd2))
        self.D = D
        rho, c = rho_cells.astype(ft), c_cells.astype(ft)
        if reference_speed_of_sound is not None:
            self.c0 = float(reference_speed_of_sound)
        elif np.ndim(speed_of_sound) == 0:
            self.c0 = float(speed_of_sound)
        elif bd1.shape[0]:
            self.c0 = float(c_cells[bd1[:, 0]].mean())
        td = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        self.cell_coeff1 = td(1.0 / rho / c / c)  # :336
        self.cell_coeff2 = td(-1.0 / rho)  # :337
        self.facet_coeff1 = td((1.0 / rho[bd1[:, 0]]).astype(ft))  # :339-341
        self.facet_coeff2 = td((-1.0 / rho[bd2[:, 0]] / c[bd2[:, 0]]).astype(ft))  # :343-345
        self.G, self.detJ, self.dofmap = G_d, detJ_d, td(mesh.dofmap)
        self.detJ_f1, self.detJ_f2 = dF1_d, dF2_d
        self.fdm1, self.fdm2 = td(mesh.facet_dofmap(bd1)), td(mesh.facet_dofmap(bd2))
        self.nlocal, self.ndofs = mesh.nlocal, mesh.ndofs

        # ---- operators --------------------------------------------------------------------------
        # affine cells (every box mesh of the reference's demos): opt into the constant-G fast path
        # after checking the geometry factors really are affine ("auto"), or on request / never
        w3 = tensor_weights_3d(gll_points_weights(P)[1])
        self.affine = bool(affine) if affine != "auto" else ops.is_affine_geometry(
            self.G, w3, rtol=1e-11 if ft == np.float64 else 1e-5)
        self.stiff = ops.stiffness_operator(P, D.flatten(), ft, affine_weights=w3 if self.affine else None)
        # opt-in for non-affine (trilinear) cells: G formed in the kernel from the cell vertices; the
        # per-cell argument of the stiffness apply is then the cell's vertex ids instead of G
        if in_kernel_geometry == "auto":
            in_kernel_geometry = P >= 3  # below that G is not the dominant stream and the kernel form loses
        self.in_kernel_geometry = bool(in_kernel_geometry) and not self.affine
        self.G_array = self.G  # the reference's geometric-factor array (None once dropped)
        if self.in_kernel_geometry:
            pts1, wts1 = gll_points_weights(P)
            self.x_dofs = td(mesh.x_dofs)
            self.stiff = ops.stiffness_operator(P, D.flatten(), ft, geometry=(self.x_dofs, mesh.x_g, pts1, wts1))
            self.G = self.x_dofs  # x_dofs rows travel in the G position (cell sub-ranges slice them)
            if not keep_G:
                self.G_array = None  # 6 n^3 values per cell nobody reads any more (config 3: 945 MB)
                del G_d
        self.mass_cell = ops.mass_operator(n**3, ft)
        self.mass_facet = ops.mass_operator(n * n, ft)
        self.axpy = ops.axpy(self.ndofs)

        # ---- halo ---------------------------------------------------------------------------------
        self.halo = None
        if comm is not None and comm.size > 1:
            from .scatterer import HaloApply, scatter_forward

            # halo_plan = (owners_data, ghosts_data) computed elsewhere (a host that drives several ranks from
            # one process has no index exchange to run); default: exchanged over ``comm`` now
            self.halo = HaloApply(mesh, self.stiff, comm, ft, overlap=overlap, kernels=halo_kernels, plan=halo_plan)
            self.fwd_v = scatter_forward(comm, self.halo.owners_data, self.halo.ghosts_data, mesh.nlocal, ft, halo_kernels)
        # the facet mass applies of a stage are ``boundary_terms`` of HaloApply: in its concurrent schedule they run on the
        # communicator's stream WHILE the interior stiffness launch adds into the same b with float atomics (interior cells
        # on the x = 0 / x = L faces touch exactly the facet dofs).  The atomic-free gather kernel's plain load + store would
        # lose those adds: next to a halo the facet operator is the float-atomic twin.
        self._mass_facet_stage = self.mass_facet if self.halo is None else self.halo.concurrent_safe(self.mass_facet)

        z = lambda: torch.zeros(self.ndofs, dtype=self.tdt, device=dev)  # noqa: E731
        self.u, self.v, self.u0, self.v0 = z(), z(), z(), z()
        self.un, self.vn, self.ku, self.kv = z(), z(), z(), z()
        self.u_n, self.v_n, self.g, self.b, self.m = z(), z(), z(), z(), z()
        self.fc1_work = torch.zeros_like(self.facet_coeff1)

        # ---- lumped mass: m = M(1/(rho c^2)) 1, reverse-scattered (:421-428) ---------------------
        ops.fill(1.0, self.g)
        # the default operator (atomic-free, bitwise reproducible): nothing else adds into m while it runs -- the reverse scatter
        # of m follows it in stream order
        self.mass_cell(self.g, self.cell_coeff1, self.m, self.detJ, self.dofmap)
        self.minv = z()
        # the reverse scatter of m: now, or (several ranks driven from one process: every rank must have
        # posted before any completes) by the driver through setup_schedule()
        self._setup = self.setup_schedule()
        if not defer_setup_exchange:
            for _ in self._setup:
                pass
        # g stays 1 for the fused path (source enters through scaled facet constants)

    def setup_schedule(self):
        """Generator: post the set-up exchange (reverse scatter of the lumped mass), yield, complete it and
        form 1/m.  One rank per process exhausts it in the constructor."""
        if self.halo is not None:
            wk = self.halo.rev.begin(self.m)
            yield "reverse"
            self.halo.rev.end(self.m, wk)
        ops.fill(1.0, self.minv)
        ops.pointwise_divide(self.minv, self.m, self.minv)  # owned entries are what the fused kernel reads

    # ------------------------------------------------------------------------------------------
    def init(self):
        """u = v = 0 (cuda/demo_linear_box.py:434-435)."""
        for t in (self.u, self.v, self.ku, self.kv):
            ops.fill(0.0, t)

    def source_value(self, t):
        """Window x p0 w0 / c0 x cos(w0 t) (cuda/demo_linear_box.py:515-530)."""
        T, alpha = 1.0 / self.f0, 4.0
        window = 0.5 * (1.0 - np.cos(self.f0 * np.pi * t / alpha)) if t < T * alpha else 1.0
        return window * self.p0 * self.w0 / self.c0 * np.cos(self.w0 * t)

    # -- reference launch sequence ----------------------------------------------------------------
    def _stage_reference(self, i, t, dt):
        copy, fill, axpy = ops.copy, ops.fill, self.axpy
        copy(self.u0, self.un)
        copy(self.v0, self.vn)
        axpy(A_RUNGE[i] * dt, self.ku, self.un)
        axpy(A_RUNGE[i] * dt, self.kv, self.vn)
        tn = t + C_RUNGE[i] * dt
        copy(self.vn, self.ku)  # f0
        fill(self.source_value(tn if self.source_time == "tn" else t), self.g)
        copy(self.un, self.u_n)
        copy(self.vn, self.v_n)
        fill(0.0, self.b)

        def facets():
            self._mass_facet_stage(self.g, self.facet_coeff1, self.b, self.detJ_f1, self.fdm1)
            self._mass_facet_stage(self.v_n, self.facet_coeff2, self.b, self.detJ_f2, self.fdm2)

        if self.halo is None:
            self.stiff(self.u_n, self.cell_coeff2, self.b, self.G, self.dofmap)
            facets()
        else:
            yield from self.halo.apply_schedule(self.u_n, self.cell_coeff2, self.b, self.G, self.dofmap,
                                                extra_forward=[(self.fwd_v, self.v_n)], boundary_terms=facets)
        ops.pointwise_divide(self.b, self.m, self.kv)
        axpy(B_RUNGE[i] * dt, self.ku, self.u)
        axpy(B_RUNGE[i] * dt, self.kv, self.v)

    # -- fused --------------------------------------------------------------------------------------
    def _rk4_stage_kernel(self, bw, aw, new_step):
        fn = getattr(_lib.load(), f"fus_rk4_stage_{_lib.suffix(self.tdt)}")
        _lib.check(
            fn(float(bw), float(aw), int(new_step), self.minv.data_ptr(), self.b.data_ptr(), self.u.data_ptr(),
               self.v.data_ptr(), self.u0.data_ptr(), self.v0.data_ptr(), self.ku.data_ptr(), self.un.data_ptr(),
               self.nlocal, self.ndofs, _lib.stream_ptr()),
            "fus_rk4_stage",
        )

    def _operator_fused(self, tn_or_t, u_n=None, v_n=None, scalars=None):
        """b += K(c2) u_n + facet terms; (u_n, v_n) default to the stage buffers (un, ku == v_n); the
        first stage of a step passes (u0, v0) themselves.  ``scalars``: device tensor the source value is read
        from instead of being evaluated at ``tn_or_t`` (graph capture)."""
        u_n = self.un if u_n is None else u_n
        v_n = self.ku if v_n is None else v_n
        gval = 0.0 if scalars is not None else self.source_value(tn_or_t)

        def facets():  # M_f1(g c1) 1 + M_f2(c2) v_n in one launch (the reference fills g into a vector)
            ops.facet_terms(self.b, (self.facet_coeff1, gval, None, 0.0, self.detJ_f1, self.fdm1),
                            (v_n, self.facet_coeff2, self.detJ_f2, self.fdm2), scalars=scalars)

        if self.halo is None:
            self.stiff(u_n, self.cell_coeff2, self.b, self.G, self.dofmap)
            facets()
        else:
            yield from self.halo.apply_schedule(u_n, self.cell_coeff2, self.b, self.G, self.dofmap,
                                                extra_forward=[(self.fwd_v, v_n)], boundary_terms=facets)

    def _stage_args(self, i, dt):
        """``(bw, aw, kind)`` of the vector pass after stage ``i`` (csrc/rk4.hpp).  Default: the LEAN set 4, 5, 6, 7 with bw = b_runge[0] dt,
        aw = a_runge[1] dt in all four passes (u's accumulator runs one pass ahead, 34 instead of 41 vector touches per linear step, 46
        instead of 52 per Westervelt step; v differs from the reference's sequence in the rounding of one term); ``lean_stages = False``
        (FUS_RK4_LEAN=0): kinds 2, 0, 0, 3, the reference's arithmetic operation for operation."""
        if self.lean_stages:
            return B_RUNGE[0] * dt, A_RUNGE[1] * dt, 4 + i
        last = i == 3
        return B_RUNGE[i] * dt, 0.0 if last else A_RUNGE[i + 1] * dt, 3 if last else (2 if i == 0 else 0)

    def rk4(self, start_time, final_time, dt, max_steps=None):
        """Advance from ``start_time`` to ``final_time`` (cuda/demo_linear_box.py:487-566).
        Returns ``(t, steps)``."""
        gen = self.rk4_schedule(start_time, final_time, dt, max_steps)
        while True:
            try:
                next(gen)
            except StopIteration as done:
                # N > 1: every device-side wait of the exchange is bounded, so a late or dead neighbour cannot hang this
                # rank -- it must not hand back a field computed from stale ghosts either (the reference would block in
                # MPI Waitall, cuda/scatterer.py:175): raise.  One synchronisation per rk4() call.
                self.check_halo_health("LinearSpectral3D.rk4")
                return done.value

    def check_halo_health(self, what="halo exchange"):
        if self.halo is not None:
            self.halo.check_health(what)

    def rk4_schedule(self, start_time, final_time, dt, max_steps=None):
        """``rk4`` as a generator that yields whenever this rank has posted halo exchanges (see
        ``HaloApply.schedule``); its return value is ``(t, steps)``.  A driver that advances several ranks' generators
        itself calls ``check_halo_health()`` when they are exhausted (``rk4`` does)."""
        t, step = float(start_time), 0
        tf = float(final_time)
        if self.fused:
            # between steps the solution lives in (u0, v0): they are the first stage's inputs as they
            # stand, and the last stage writes the new solution straight into them (stage kinds 2, 0, 0,
            # 3 of csrc/rk4.hpp: 41 instead of 48 vector touches per step)
            ops.fill(0.0, self.b)
            ops.copy(self.u, self.u0)
            ops.copy(self.v, self.v0)
        while t < tf and (max_steps is None or step < max_steps):
            dt = min(dt, tf - t)
            if self.fused:
                for i in range(4):
                    tn = t + C_RUNGE[i] * dt
                    if i == 0:
                        yield from self._operator_fused(tn if self.source_time == "tn" else t, self.u0, self.v0)
                    else:
                        yield from self._operator_fused(tn if self.source_time == "tn" else t)
                    self._rk4_stage_kernel(*self._stage_args(i, dt))
            else:
                ops.copy(self.u, self.u0)
                ops.copy(self.v, self.v0)
                for i in range(4):
                    yield from self._stage_reference(i, t, dt)
            t += dt
            step += 1
        if self.fused:
            ops.copy(self.u0, self.u)
            ops.copy(self.v0, self.v)
        return t, step

    # -- hipGraph replay (launch-bound meshes): step_graph.StepGraphMixin.rk4_graph ---------------------
    def _graph_state(self):
        return (self.u, self.v, self.u0, self.v0, self.ku, self.un, self.b)

    def _graph_scalars(self, t):
        return (self.source_value(t), 0.0)

    def _graph_enter(self):
        ops.fill(0.0, self.b)
        ops.copy(self.u, self.u0)
        ops.copy(self.v, self.v0)

    def _graph_exit(self):
        ops.copy(self.u0, self.u)
        ops.copy(self.v0, self.v)

    def _graph_step_body(self, dt):
        """The 8 launches of one fused RK4 step with every argument a fixed pointer or a constant: the four
        source values are read from ``self._scal[i]``."""
        for i in range(4):
            first, last = i == 0, i == 3
            for _ in self._operator_fused(None, self.u0 if first else None, self.v0 if first else None, scalars=self._scal[i]):
                pass
            self._rk4_stage_kernel(*self._stage_args(i, dt))

    def u_sol(self, with_ghosts=False):
        """Owned part of the pressure field on the host; ``with_ghosts``: the whole local vector after a forward scatter
        (``scatter_fwd(u_n_d); u_n_d.copy_to_host(u_n)``, cuda/demo_linear_box.py:568-570 -- what point evaluation needs)."""
        if not with_ghosts:
            return self.u[: self.nlocal].detach().cpu().numpy()
        if self.halo is not None:
            self.halo.fwd(self.u)
            torch.cuda.synchronize()
            self.check_halo_health("LinearSpectral3D.u_sol(with_ghosts=True)")
        return self.u.detach().cpu().numpy()

    def v_sol(self):
        return self.v[: self.nlocal].detach().cpu().numpy()
