"""
Westervelt (nonlinear) acoustic wave solver on the synthetic mesh -- the caller of BASELINE
config 5 (cuda/demo_nonlinear_bowl.py / demo_nonlinear_box.py): per RK4 stage two stiffness
applies (u_n with -1/rho, v_n with -delta/(rho c^2)), a solution-dependent lumped mass
m = m0 + M(-2 beta/(rho^2 c^4)) u_n, the quadratic term M(2 beta/(rho^2 c^4)) v_n^2 and the
source / absorbing boundary-facet terms (cuda/demo_nonlinear_bowl.py:357-374, 458-475, 540-650).

The reference's transducer mesh (H131/mesh.xdmf) is not in its repository; the geometry here is
the synthetic box, optionally warped into non-affine trilinear cells (``BoxMesh(warp=...)``), with
the source on x = 0 and the absorbing condition on x = L as in cuda/demo_nonlinear_box.py.
The stage follows the reference's launch sequence through the reference-compatible operators.
"""

from __future__ import annotations

import os

import numpy as np
import torch

from . import _lib
from . import operators as ops
from .linear_solver import A_RUNGE, B_RUNGE, C_RUNGE
from .linear_solver import device_geometry
from .step_graph import StepGraphMixin


def compute_diffusivity_of_sound(frequency, speed, attenuationdB):
    """cuda/utils.py:157-162."""
    attenuationNp = attenuationdB / 20 * np.log(10)
    return 2 * attenuationNp * speed**3 / frequency / frequency


class WesterveltSpectral3D(StepGraphMixin):
    def __init__(self, mesh, float_type=np.float64, speed_of_sound=1480.0, density=1000.0,
                 source_frequency=1.1e6, source_amplitude=None, nonlinear_coefficient=3.5,
                 attenuation_coefficient_dB=0.2, comm=None, source_time="tn", overlap=True, fused=False,
                 in_kernel_geometry="auto", uniform_ratio="auto", halo_plan=None, defer_setup_exchange=False,
                 reference_speed_of_sound=None, reference_density=None, keep_G=False):
        """``speed_of_sound``, ``density``, ``nonlinear_coefficient``, ``attenuation_coefficient_dB``: scalars, or one value per
        cell in the caller's cell order (the DG0 material arrays of cuda/demo_nonlinear_bowl.py:166-178 -- water / skull / ...).
        ``reference_speed_of_sound`` / ``reference_density``: the scalars of the source term and of the default source amplitude
        (the reference uses those of the coupling medium); default: the scalars given, or the means over the source-facet cells.
        ``uniform_ratio``: ``True`` opts into the single-gather cell pass where c4 / c3 = delta / c^2 is uniform (any homogeneous
        medium); the default is the two-gather pass for every medium (faster since round 5, and what a heterogeneous medium needs anyway).
        ``in_kernel_geometry``: ``"auto"`` (default) -- the fused stage of degree >= 3 forms G in the cell kernel from the 8
        vertices of each (trilinear) cell and the G array is dropped unless ``keep_G``; ``False``: the reference's G stream;
        the reference launch sequence (``fused=False``) always reads G."""
        from .linear_solver import per_cell

        if comm is not None:  # an MPI.Comm (the reference's comm = MPI.COMM_WORLD) becomes the bootstrap of a NativeComm
            from .scatterer import as_comm

            comm = as_comm(comm)
        self.mesh, self.P = mesh, mesh.P
        ft = np.dtype(float_type)
        self.tdt_np = ft
        self.tdt = _lib.torch_dtype(ft)
        c_cells, rho_cells = per_cell(speed_of_sound, mesh, "speed_of_sound"), per_cell(density, mesh, "density")
        beta_cells = per_cell(nonlinear_coefficient, mesh, "nonlinear_coefficient")
        att_cells = per_cell(attenuation_coefficient_dB, mesh, "attenuation_coefficient_dB")
        self.f0 = float(source_frequency)
        self.w0 = 2 * np.pi * self.f0
        src = mesh.boundary_facets([getattr(mesh, "source_tag", 2)])
        pick = (lambda a: float(a[src[:, 0]].mean())) if src.shape[0] else (lambda a: float(a.mean()))
        self.c0 = float(reference_speed_of_sound) if reference_speed_of_sound is not None else (
            float(speed_of_sound) if np.ndim(speed_of_sound) == 0 else pick(c_cells))
        self.rho0 = float(reference_density) if reference_density is not None else (
            float(density) if np.ndim(density) == 0 else pick(rho_cells))
        self.p0 = float(source_amplitude) if source_amplitude is not None else self.rho0 * self.c0 * 0.38557513826589934
        self.beta = float(beta_cells.mean())
        delta_cells = compute_diffusivity_of_sound(self.w0, c_cells, att_cells)
        self.delta = float(delta_cells.mean())
        self.source_time = source_time
        self.fused = bool(fused)
        self.lean_stages = os.environ.get("FUS_RK4_LEAN", "1") != "0"  # the fused stage's vector pass: kinds 4-7 of csrc/rk4.hpp (_stage_args)
        P, n = self.P, self.P + 1
        dev = torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        nc = mesh.ncells
        # tagged facet sets: source / absorbing (a structured box: its x = 0 / x = L faces; dolfinx_adaptor.ArrayMesh: facet tags)
        bd1, bd2 = mesh.boundary_facets([getattr(mesh, "source_tag", 2)]), mesh.boundary_facets([getattr(mesh, "absorbing_tag", 3)])
        D, G_d, detJ_d, (dF1_d, dF2_d) = device_geometry(mesh, P, ft, dev, (bd1, bd2))
        rho, c, beta, delta = rho_cells, c_cells, beta_cells, delta_cells
        td = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=ft))).to(dev)  # noqa: E731
        # cuda/demo_nonlinear_bowl.py:357-374
        self.cc1 = td(1.0 / rho / c / c)
        self.cc2 = td(-2.0 * beta / rho / rho / c**4)
        self.cc3 = td(-1.0 / rho)
        self.cc4 = td(-delta / rho / c / c)
        self.cc5 = td(2.0 * beta / rho / rho / c**4)
        c1, c2 = bd1[:, 0], bd2[:, 0]
        self.fc1_1 = td(1.0 / rho[c1])
        self.fc2_1 = td(delta[c1] / rho[c1] / c[c1] ** 2)
        self.fc1_2 = td(delta[c2] / rho[c2] / c[c2] ** 3)
        self.fc2_2 = td(-1.0 / rho[c2] / c[c2])
        self.G, self.detJ = G_d, detJ_d
        self.dofmap = torch.from_numpy(mesh.dofmap).to(dev)
        self.dF1, self.dF2 = dF1_d, dF2_d
        self.fdm1 = torch.from_numpy(mesh.facet_dofmap(bd1)).to(dev)
        self.fdm2 = torch.from_numpy(mesh.facet_dofmap(bd2)).to(dev)
        self.nlocal, self.ndofs = mesh.nlocal, mesh.ndofs
        self.stiff = ops.stiffness_operator(P, D.flatten(), ft)
        # detJ never changes in the life of a solver: the reference-sequence stage applies the cell mass operator twice per stage
        # with it (cuda/demo_nonlinear_bowl.py:612-616, 630-632) -- streamed from a row-ordered copy instead of gathered
        self.mass_cell = ops.mass_operator(n**3, ft, static_detJ=not bool(fused))
        self.mass_facet = ops.mass_operator(n * n, ft)
        self.axpy = ops.axpy(self.ndofs)
        self.halo = None
        if comm is not None and comm.size > 1:
            from .scatterer import HaloApply, scatter_forward

            self.halo = HaloApply(mesh, self.stiff, comm, ft, overlap=overlap, plan=halo_plan)
            mk = lambda: scatter_forward(comm, self.halo.owners_data, self.halo.ghosts_data, mesh.nlocal, ft)  # noqa: E731
            self.fwd_u, self.fwd_v, self.fwd_w = self.halo.fwd, mk(), mk()
            from .scatterer import scatter_reverse

            self.rev_m = scatter_reverse(comm, self.halo.owners_data, self.halo.ghosts_data, mesh.nlocal, ft)
        z = lambda: torch.zeros(self.ndofs, dtype=self.tdt, device=dev)  # noqa: E731
        (self.u, self.v, self.u0, self.v0, self.un, self.vn, self.ku, self.kv, self.u_n, self.v_n, self.w_n,
         self.g, self.dg, self.b, self.m, self.m0) = (z() for _ in range(16))
        # steady part of the lumped mass (:458-475)
        ops.fill(1.0, self.g)
        # set-up applies on the default operator (atomic-free, bitwise reproducible), like the reference-sequence stage's two
        # cell mass applies: each runs alone on this stream, the reverse scatters follow in stream order
        self.mass_cell(self.g, self.cc1, self.m0, self.detJ, self.dofmap)
        self.mass_facet(self.g, self.fc1_2, self.m0, self.dF2, self.fdm2)

        self.cell_fused = ops.westervelt_cell_operator(P, D.flatten(), ft)
        # fused mode: with GLL collocation the mass operator is diagonal, M(c) x = diag(M(c) 1) x, so the
        # stage's two cell mass applies (M(c2) u_n for the lumped mass, M(c5) v_n^2 for the right-hand side,
        # cuda/demo_nonlinear_bowl.py:612-616,630-632) are pointwise products with two diagonals assembled
        # once, like m0; the cell pass is then the stiffness part alone and m needs no reverse scatter
        self.w2, self.w5 = z(), z()
        self.mass_cell(self.g, self.cc2, self.w2, self.detJ, self.dofmap)  # g == 1 here
        self.mass_cell(self.g, self.cc5, self.w5, self.detJ, self.dofmap)
        # the reverse scatters of the three assembled diagonals (m0, w2, w5), as one grouped exchange: now, or
        # by a driver that runs several ranks from one process (setup_schedule, see LinearSpectral3D)
        # the third reverse closure of the set-up exchange is built HERE, with the others: building a closure is a collective
        # step of the PEER transport (arena handles), and ranks driven from one process must all have built theirs before
        # any of them exchanges
        self._rev_w5 = None
        if self.halo is not None:
            from .scatterer import scatter_reverse

            self._rev_w5 = scatter_reverse(self.halo.comm, self.halo.owners_data, self.halo.ghosts_data, self.nlocal, self.tdt_np)
        self._setup = self.setup_schedule()
        if not defer_setup_exchange:
            for _ in self._setup:
                pass
        # uniform ratio c4 / c3 (= delta / c^2: every homogeneous medium): K(c3) u + K(c4) v = K(c3)(u + kappa v),
        # so the cell pass CAN be one plain stiffness apply on w = u_n + kappa v_n, which the vector kernel writes (uniform_ratio=True)
        ratio = self.cc4 / self.cc3
        kmin, kmax = float(ratio.min().item()), float(ratio.max().item())
        # the decision (and kappa itself) must be the SAME on every rank -- a rank in single-gather mode forward-scatters
        # w where a neighbour in two-gather mode expects u_n, with matching counts, so nothing would hang and the result
        # would be silently wrong: min / max over all ranks.  cc3 / cc4 must not be edited after construction.
        if comm is not None and getattr(comm, "size", 1) > 1 and getattr(comm, "_world_id", None) is None:
            import torch.distributed as dist

            if hasattr(comm, "allgather_floats"):  # NativeComm: over its bootstrap (torch.distributed or MPI)
                every = comm.allgather_floats([kmin, kmax])
                kmin, kmax = float(every[:, 0].min()), float(every[:, 1].max())
            elif dist.is_available() and dist.is_initialized():
                on_gpu = dist.get_backend() == "nccl"
                t = torch.tensor([-kmin, kmax], dtype=torch.float64, device=self.dev if on_gpu else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                kmin, kmax = -float(t[0].item()), float(t[1].item())
        self.kappa = kmin if abs(kmax - kmin) <= 1e-14 * max(abs(kmin), abs(kmax), 1e-300) else None
        # Which form: since the vector pass streams with non-temporal accesses (round 4) the TWO-gather cell pass is the faster one --
        # the single-gather form pays for writing w in the vector pass and re-reading it: P = 6, 36^3 cells, paired: 1.285 against
        # 1.458 ms per step with in-kernel geometry, 1.538 against 1.593 with the G array (profiles/r05i_ab_westervelt_gathers.log).
        # "auto" / False: two gathers; True: the single-gather form where the medium allows it.
        if uniform_ratio is not True:
            self.kappa = None
        self.w = z() if self.kappa is not None else None
        # opt-in (fused mode): G and detJ formed in the cell kernel from the vertices -- the cells of
        # the reference's meshes are trilinear (P1 geometry, cuda/demo_nonlinear_bowl.py:317)
        if in_kernel_geometry == "auto":
            in_kernel_geometry = self.fused and P >= 3
        self.in_kernel_geometry = bool(in_kernel_geometry)
        if self.in_kernel_geometry and self.fused and not keep_G:
            self.G = None  # the fused stage does not read it (P = 6, 36^3 cells: 768 MB)
        if self.in_kernel_geometry:
            from .gll import tabulate_1d

            pts, wts, _ = tabulate_1d(P, ft)
            self.x_dofs = torch.from_numpy(np.ascontiguousarray(mesh.x_dofs)).to(dev)
            self.cell_fused_geom = ops.westervelt_cell_operator(P, D.flatten(), ft, geometry=(mesh.x_g, pts, wts))
            self.stiff_geom = ops.stiffness_operator(P, D.flatten(), ft, geometry=(self.x_dofs, mesh.x_g, pts, wts))
        self.fc_src = torch.zeros_like(self.fc1_1)  # per-stage source-facet constants (fused mode)

    def setup_schedule(self):
        if self.halo is not None:
            from .scatterer import begin_all

            self._rev_setup = [self.halo.rev, self.rev_m, self._rev_w5]
            pending = begin_all(zip(self._rev_setup, (self.m0, self.w2, self.w5)))
            yield "reverse"
            for sc, vec, wk in pending:
                sc.end(vec, wk)

    def init(self):
        for t in (self.u, self.v, self.ku, self.kv):
            ops.fill(0.0, t)

    # -- fused stage: one cell pass + one vector pass ------------------------------------------------
    def _stage_vector_kernel(self, bw, aw, new_step):
        fn = getattr(_lib.load(), f"fus_rk4_stage_nl2_{_lib.suffix(self.tdt)}")
        _lib.check(
            fn(float(bw), float(aw), int(new_step), self.m0.data_ptr(), self.w2.data_ptr(), self.w5.data_ptr(),
               self.b.data_ptr(), self.u.data_ptr(), self.v.data_ptr(), self.u0.data_ptr(), self.v0.data_ptr(),
               self.ku.data_ptr(), self.un.data_ptr(), float(self.kappa or 0.0),
               self.w.data_ptr() if self.w is not None else None, self.nlocal, self.ndofs, _lib.stream_ptr()),
            "fus_rk4_stage_nl2",
        )

    def _operator_fused(self, ts, u_n=None, v_n=None, scalars=None):
        u_n = self.un if u_n is None else u_n
        v_n = self.ku if v_n is None else v_n  # ku == v_n
        gv, dgv = (0.0, 0.0) if scalars is not None else self.source_values(ts)  # scalars: (g, dg) in device memory

        single = self.kappa is not None  # one gather: the cell pass is K(c3) w, w = u_n + kappa v_n
        w_n = self.w

        def cells(c3, c4, G_, dm_):
            if single:
                self.stiff(w_n, c3, self.b, G_, dm_)
            else:
                self.cell_fused.stiffness_only(u_n, v_n, c3, c4, self.b, G_, dm_)

        def facets():  # M_f1(fc1_1 g + fc2_1 dg) 1 + M_f2(fc2_2) v_n in one launch
            ops.facet_terms(self.b, (self.fc1_1, gv, self.fc2_1, dgv, self.dF1, self.fdm1), (v_n, self.fc2_2, self.dF2, self.fdm2),
                            scalars=scalars)

        percell = (self.cc3, self.cc4, self.G, self.dofmap)
        if self.in_kernel_geometry:
            def cells(c3, c4, xd_, dm_):  # noqa: F811
                if single:
                    self.stiff_geom(w_n, c3, self.b, xd_, dm_)  # x_dofs rows travel in the G position
                else:
                    self.cell_fused_geom.stiffness_only(u_n, v_n, c3, c4, self.b, xd_, dm_)

            percell = (self.cc3, self.cc4, self.x_dofs, self.dofmap)
        if self.halo is None:
            cells(*percell)
            facets()
        else:
            yield from self.halo.schedule(cells, percell, [(self.fwd_u, w_n if single else u_n), (self.fwd_v, v_n)],
                                          [(self.halo.rev, self.b)], facets)

    def source_values(self, t):
        """g and dg/dt (cuda/demo_nonlinear_bowl.py:560-595)."""
        T, alpha = 1.0 / self.f0, 4.0
        if t < T * alpha:
            window = 0.5 * (1.0 - np.cos(self.f0 * np.pi * t / alpha))
            dwindow = 0.5 * np.pi * self.f0 / alpha * np.sin(self.f0 * np.pi * t / alpha)
        else:
            window, dwindow = 1.0, 0.0
        a = 2.0 * self.p0 * self.w0 / self.c0
        g = window * a * np.cos(self.w0 * t)
        dg = dwindow * a * np.cos(self.w0 * t) - window * a * self.w0 * np.sin(self.w0 * t)
        return g, dg

    def _stage(self, i, t, dt):
        copy, fill, axpy = ops.copy, ops.fill, self.axpy
        copy(self.u0, self.un)
        copy(self.v0, self.vn)
        axpy(A_RUNGE[i] * dt, self.ku, self.un)
        axpy(A_RUNGE[i] * dt, self.kv, self.vn)
        tn = t + C_RUNGE[i] * dt
        copy(self.vn, self.ku)
        gv, dgv = self.source_values(tn if self.source_time == "tn" else t)
        fill(gv, self.g)
        fill(dgv, self.dg)
        copy(self.un, self.u_n)
        copy(self.vn, self.v_n)
        ops.square(self.vn, self.w_n)
        if self.halo is not None:
            self.fwd_u(self.u_n)
            self.fwd_v(self.v_n)
            self.fwd_w(self.w_n)
        # unsteady lumped mass
        fill(0.0, self.m)
        self.mass_cell(self.u_n, self.cc2, self.m, self.detJ, self.dofmap)
        if self.halo is not None:
            self.halo.rev(self.m)
        axpy(1.0, self.m0, self.m)
        # right-hand side
        fill(0.0, self.b)
        self.stiff(self.u_n, self.cc3, self.b, self.G, self.dofmap)
        self.stiff(self.v_n, self.cc4, self.b, self.G, self.dofmap)
        self.mass_cell(self.w_n, self.cc5, self.b, self.detJ, self.dofmap)
        self.mass_facet(self.g, self.fc1_1, self.b, self.dF1, self.fdm1)
        self.mass_facet(self.dg, self.fc2_1, self.b, self.dF1, self.fdm1)
        self.mass_facet(self.v_n, self.fc2_2, self.b, self.dF2, self.fdm2)
        if self.halo is not None:
            self.halo.rev(self.b)
        ops.pointwise_divide(self.b, self.m, self.kv)
        axpy(B_RUNGE[i] * dt, self.ku, self.u)
        axpy(B_RUNGE[i] * dt, self.kv, self.v)

    # -- hipGraph replay (launch-bound meshes): step_graph.StepGraphMixin.rk4_graph ---------------------
    def _graph_state(self):
        return (self.u, self.v, self.u0, self.v0, self.ku, self.un, self.b) + ((self.w,) if self.w is not None else ())

    def _graph_scalars(self, t):
        return self.source_values(t)

    def _graph_enter(self):
        ops.fill(1.0, self.g)
        ops.fill(0.0, self.b)
        ops.copy(self.u, self.u0)
        ops.copy(self.v, self.v0)
        if self.kappa is not None:
            ops.copy(self.u0, self.w)
            self.axpy(self.kappa, self.v0, self.w)

    def _graph_exit(self):
        ops.copy(self.u0, self.u)
        ops.copy(self.v0, self.v)

    def _graph_step_body(self, dt):
        for i in range(4):
            first, last = i == 0, i == 3
            for _ in self._operator_fused(None, self.u0 if first else None, self.v0 if first else None, scalars=self._scal[i]):
                pass
            self._stage_vector_kernel(*self._stage_args(i, dt))

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
        gen = self.rk4_schedule(start_time, final_time, dt, max_steps)
        while True:
            try:
                next(gen)
            except StopIteration as done:
                self.check_halo_health("WesterveltSpectral3D.rk4")  # a failed exchange is an error, not a field (see LinearSpectral3D.rk4)
                return done.value

    def check_halo_health(self, what="halo exchange"):
        if self.halo is not None:
            self.halo.check_health(what)

    def rk4_schedule(self, start_time, final_time, dt, max_steps=None):
        """``rk4`` as a generator that yields whenever this rank has posted halo exchanges; returns ``(t, steps)``."""
        t, step, tf = float(start_time), 0, float(final_time)
        if self.fused:
            ops.fill(1.0, self.g)  # source enters through scaled facet constants
            ops.fill(0.0, self.b)
            ops.copy(self.u, self.u0)  # between steps the solution lives in (u0, v0): stage kinds 2, 0, 0, 3
            ops.copy(self.v, self.v0)
            if self.kappa is not None:
                ops.copy(self.u0, self.w)
                self.axpy(self.kappa, self.v0, self.w)
        while t < tf and (max_steps is None or step < max_steps):
            dt = min(dt, tf - t)
            if self.fused:
                for i in range(4):
                    tn = t + C_RUNGE[i] * dt
                    if i == 0:
                        yield from self._operator_fused(tn if self.source_time == "tn" else t, self.u0, self.v0)
                    else:
                        yield from self._operator_fused(tn if self.source_time == "tn" else t)
                    self._stage_vector_kernel(*self._stage_args(i, dt))
            else:
                ops.copy(self.u, self.u0)
                ops.copy(self.v, self.v0)
                for i in range(4):
                    self._stage(i, t, dt)
            t += dt
            step += 1
        if self.fused:
            ops.copy(self.u0, self.u)
            ops.copy(self.v0, self.v)
        return t, step

    def u_sol(self):
        return self.u[: self.nlocal].detach().cpu().numpy()

    def v_sol(self):
        return self.v[: self.nlocal].detach().cpu().numpy()
