#!/usr/bin/env python3
"""
Generate the golden input/output vectors under tests/golden/*.npz by RUNNING
THE REFERENCE ITSELF (numba-cpu/{operators,sum_factorisation,precompute,
scatterer}.py imported from /root/reference) on small synthetic meshes.

Run in the build container only (the reference never travels to the GPU box):

    python tests/golden/generate_golden.py

``numba`` and ``mpi4py`` are not installed here (plain ModuleNotFoundError);
the reference uses ``numba.njit`` purely as a decorator and ``mpi4py`` only for
``comm.Isend/Irecv`` inside the scatter closures, so two in-memory stand-in
modules (identity ``njit``; an in-process ``FakeComm``) let the reference code
execute unmodified as plain Python/numpy.  Nothing from the reference is copied:
the committed artefacts are data only (inputs + the reference's outputs).

Inputs (meshes, GLL tables, P1 geometry gradients) come from this repo's host
plumbing; every array the reference consumed is stored next to what it
produced, so the fixtures are self-contained.
"""

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = os.environ.get("FUS_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)


def _install_stubs():
    nb = types.ModuleType("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda f: f

    nb.njit = njit
    nb.types = types.SimpleNamespace(Array=object)
    sys.modules["numba"] = nb

    mpi = types.ModuleType("mpi4py")
    MPI = types.ModuleType("mpi4py.MPI")

    class Request:
        @staticmethod
        def Waitall(reqs):
            for r in reqs:
                if r is not None:
                    r()

    class Comm:  # only for annotations
        pass

    MPI.Request = Request
    MPI.Comm = Comm
    MPI.COMM_WORLD = None  # set to the FakeComm of the simulated rank before each reference call
    mpi.MPI = MPI
    sys.modules["mpi4py"] = mpi
    sys.modules["mpi4py.MPI"] = MPI

    # cuda/utils.py imports dolfinx names it only uses in functions that are never called here
    dfx = types.ModuleType("dolfinx")
    dmesh = types.ModuleType("dolfinx.mesh")
    dmesh.Mesh = object
    dgeo = types.ModuleType("dolfinx.geometry")
    dgeo.bb_tree = dgeo.compute_collisions_points = dgeo.compute_colliding_cells = None
    dfx.mesh, dfx.geometry = dmesh, dgeo
    sys.modules["dolfinx"] = dfx
    sys.modules["dolfinx.mesh"] = dmesh
    sys.modules["dolfinx.geometry"] = dgeo


class FakeWorld:
    """In-process mailbox shared by the FakeComm of every simulated rank."""

    def __init__(self):
        self.box = {}


class FakeComm:
    def __init__(self, world, rank):
        self.world, self.rank = world, rank

    def Isend(self, buf, dest):
        self.world.box[(self.rank, int(dest))] = np.array(buf, copy=True)
        return None

    def Irecv(self, buf, source):
        def complete():
            buf[:] = self.world.box[(int(source), self.rank)]

        return complete


def main():
    _install_stubs()
    sys.path.insert(0, os.path.join(REF, "numba-cpu"))
    import operators as ref_ops  # noqa: E402  (reference)
    import precompute as ref_pre  # noqa: E402  (reference)
    import scatterer as ref_sc  # noqa: E402  (reference)

    import fusgpu_loader

    gll = fusgpu_loader.submodule("gll")
    boxmesh = fusgpu_loader.submodule("boxmesh")
    pre = fusgpu_loader.submodule("precompute")
    utils = fusgpu_loader.submodule("utils")

    def test_function(xyz):
        # numba-cpu/test_operators.py:274-279
        return 100 * np.sin(2 * np.pi * xyz[:, 0]) * np.cos(3 * np.pi * xyz[:, 1]) * np.sin(4 * np.pi * xyz[:, 2])

    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else "all"  # all | ops | scatter | plan | rk4 | rk4nl
    # --match SUBSTRING: (re)write only the ops / rk4nl fixtures whose tag contains it (round 6 added P = 5, P = 8 and the P = 3 two-rank
    # Westervelt case without touching the committed ones: ``--only ops --match ops_P5``, ``--only rk4nl --match P3``)
    match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else ""

    # ---- operator + precompute fixtures --------------------------------------
    cases = []
    for P in (2, 3, 4, 6):
        for shape in ((2, 2, 2), (3, 2, 2)):
            for perturb in (0.0, 0.16):
                for dt in (np.float64, np.float32):
                    if P == 6 and shape == (3, 2, 2):
                        continue  # keep the fixture set small
                    cases.append((P, shape, perturb, dt))
    # the degrees between / above the reference's test set that its Q map serves (numba-cpu/time_operators.py:35-45): the generic-n path
    # of the operators at an odd and at a high degree
    cases += [(5, (2, 2, 2), 0.16, np.float64), (8, (2, 2, 2), 0.16, np.float64)]
    # ... and the rest of its map, on fewer cells (the fixtures stay small): with these every degree 2 ... 10 is pinned by reference-held data
    cases += [(7, (2, 2, 1), 0.16, np.float64), (9, (2, 1, 1), 0.16, np.float64), (10, (2, 1, 1), 0.16, np.float64)]
    for P, shape, perturb, dt in (cases if only in ("all", "ops") else []):
        if match and match not in f"ops_P{P}_{shape[0]}x{shape[1]}x{shape[2]}_{'pert' if perturb else 'affine'}_{np.dtype(dt).name}":
            continue
        n = P + 1
        mesh = boxmesh.BoxMesh(P, shape, perturb=perturb, seed=7, dtype=dt)
        pts, wts, D = gll.tabulate_1d(P, dt)
        wts3 = gll.tensor_weights_3d(wts).astype(dt)
        dphi_g = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts), dt)
        nc = mesh.ncells
        detJ = np.zeros((nc, n**3), dtype=dt)
        G = np.zeros((nc, n**3, 6), dtype=dt)
        ref_pre.compute_scaled_jacobian_determinant(detJ, (mesh.x_dofs, mesh.x_g), nc, dphi_g, wts3)
        ref_pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), nc, dphi_g, wts3)
        bdata = mesh.boundary_facets()
        wts2 = gll.tensor_weights_2d(wts).astype(dt)
        dphi_f = pre.tabulate_facet_gradients(pts, dt)
        detJ_f = np.zeros((bdata.shape[0], n * n), dtype=dt)
        ref_pre.compute_boundary_facets_scaled_jacobian_determinant(
            detJ_f, (mesh.x_dofs, mesh.x_g), bdata, dphi_f, wts2
        )
        bfacet_dofmap = mesh.facet_dofmap(bdata)

        rng = np.random.default_rng(1234 + P)
        x = test_function(mesh.dof_coordinates()).astype(dt)
        cell_constants = (1.0 + 0.5 * rng.standard_normal(nc)).astype(dt)
        facet_constants = (1.0 + 0.5 * rng.standard_normal(bdata.shape[0])).astype(dt)
        y0 = rng.standard_normal(mesh.ndofs).astype(dt)  # operators accumulate INTO y

        y_stiff = y0.copy()
        ref_ops.stiffness_operator(P, D.flatten(), dt)(x, cell_constants, y_stiff, G, mesh.dofmap)
        y_mass = y0.copy()
        ref_ops.mass_operator(n**3, dt)(x, cell_constants, y_mass, detJ, mesh.dofmap)
        y_fmass = y0.copy()
        ref_ops.mass_operator(n * n, dt)(x, facet_constants, y_fmass, detJ_f, bfacet_dofmap)

        # streaming vector ops (numba-cpu/operators.py:230-300)
        va = rng.standard_normal(mesh.ndofs).astype(dt)
        vb = (2.0 + rng.random(mesh.ndofs)).astype(dt)
        alpha = dt(0.37)
        y_axpy = vb.copy()
        ref_ops.axpy(mesh.ndofs)(alpha, va, y_axpy)
        y_div = np.zeros_like(va)
        ref_ops.pointwise_divide(va, vb, y_div)

        tag = f"ops_P{P}_{shape[0]}x{shape[1]}x{shape[2]}_{'pert' if perturb else 'affine'}_{np.dtype(dt).name}"
        np.savez_compressed(
            os.path.join(HERE, tag + ".npz"),
            P=P,
            shape=np.array(shape),
            perturb=perturb,
            x_dofs=mesh.x_dofs,
            x_g=mesh.x_g,
            dofmap=mesh.dofmap,
            pts=pts,
            wts=wts,
            dphi_1d=D,
            wts3=wts3,
            wts2=wts2,
            dphi_geom=dphi_g,
            dphi_facet=dphi_f,
            boundary_data=bdata,
            bfacet_dofmap=bfacet_dofmap,
            x=x,
            cell_constants=cell_constants,
            facet_constants=facet_constants,
            y0=y0,
            ref_detJ=detJ,
            ref_G=G,
            ref_detJ_f=detJ_f,
            ref_y_stiffness=y_stiff,
            ref_y_mass=y_mass,
            ref_y_facet_mass=y_fmass,
            va=va,
            vb=vb,
            alpha=alpha,
            ref_y_axpy=y_axpy,
            ref_y_divide=y_div,
        )
        print("wrote", tag, f"ncell={nc} ndofs={mesh.ndofs}")

    # ---- scatterer fixtures: reference closures on simulated ranks ------------
    scatter_cases = ((2, (4, 2, 2), (2, 1, 1)), (3, (2, 4, 2), (1, 2, 1)), (2, (4, 4, 2), (2, 2, 1)), (2, (2, 2, 2), (2, 2, 2)))
    for P, shape, grid in (scatter_cases if only in ("all", "scatter") else []):
        R = int(np.prod(grid))
        meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r) for r in range(R)]
        od_all, gd_all = utils.compute_scatterer_data_all([m.index_map for m in meshes])
        world = FakeWorld()
        rng = np.random.default_rng(99)
        bufs = [rng.standard_normal(m.ndofs) for m in meshes]
        out = {"P": P, "shape": np.array(shape), "grid": np.array(grid)}
        rev = [b.copy() for b in bufs]
        fwd = [b.copy() for b in bufs]
        for kind, arrs, factory in (("rev", rev, ref_sc.scatter_reverse), ("fwd", fwd, ref_sc.scatter_forward)):
            world.box.clear()
            closures = [
                factory(FakeComm(world, r), od_all[r], gd_all[r], meshes[r].nlocal, np.float64) for r in range(R)
            ]
            # phase 1: every rank packs + posts sends; phase 2: completes recvs + unpacks.
            # The reference closure does both in one call; run it rank by rank with the
            # sends of ALL ranks already in the mailbox (two passes over a scratch copy).
            scratch = [a.copy() for a in arrs]
            for r in range(R):
                try:
                    closures[r](scratch[r])
                except KeyError:
                    pass  # recv from a rank that has not "sent" yet; mailbox now holds r's sends
            for r in range(R):
                closures[r](arrs[r])
        for r in range(R):
            out[f"in_{r}"] = bufs[r]
            out[f"ref_rev_{r}"] = rev[r]
            out[f"ref_fwd_{r}"] = fwd[r]
            out[f"nlocal_{r}"] = meshes[r].nlocal
        tag = f"scatter_P{P}_{shape[0]}x{shape[1]}x{shape[2]}_grid{grid[0]}x{grid[1]}x{grid[2]}"
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
        print("wrote", tag)


    # ---- halo plan fixtures: the reference's own compute_scatterer_data -------------------
    # cuda/utils.py:8-78, run rank by rank on BoxMesh index maps (which duck-type the dolfinx
    # IndexMap members it touches: size_local, num_ghosts, owners, ghosts, local_range,
    # index_to_dest_ranks()), MPI.COMM_WORLD = the simulated rank's FakeComm.
    if only in ("all", "plan"):
        import importlib.util
        import mpi4py.MPI as MPI  # the stub

        spec = importlib.util.spec_from_file_location("ref_cuda_utils", os.path.join(REF, "cuda", "utils.py"))
        ref_utils = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref_utils)
        plan_cases = [(2, (4, 2, 2), (2, 1, 1), "owner"), (3, (2, 4, 2), (1, 2, 1), "owner"), (2, (4, 4, 2), (2, 2, 1), "owner"),
                      (2, (2, 2, 2), (2, 2, 2), "owner"), (2, (4, 4, 2), (2, 2, 1), 7), (2, (4, 4, 4), (2, 2, 2), "lex"),
                      (3, (3, 3, 3), (3, 1, 1), 5)]
        for P, shape, grid, ghost_order in plan_cases:
            R = int(np.prod(grid))
            meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r, ghost_order=ghost_order) for r in range(R)]
            world = FakeWorld()
            results = [None] * R
            for final in (False, True):  # pass 1 fills the mailbox with every rank's sends
                for r in range(R):
                    MPI.COMM_WORLD = FakeComm(world, r)
                    try:
                        results[r] = ref_utils.compute_scatterer_data(meshes[r].index_map)
                    except KeyError:
                        assert not final
            out = {"P": P, "shape": np.array(shape), "grid": np.array(grid), "ghost_order": str(ghost_order)}
            for r in range(R):
                (o_idx, o_size, o_ranks), (g_idx, g_size, g_ranks) = results[r]
                cat = lambda lst: np.concatenate([np.asarray(a, dtype=np.int64) for a in lst]) if len(lst) else np.zeros(0, np.int64)  # noqa: E731
                out[f"owners_idx_{r}"] = cat(o_idx)
                out[f"owners_size_{r}"] = np.asarray(o_size, dtype=np.int64)
                out[f"unique_owners_{r}"] = np.asarray(o_ranks, dtype=np.int64)
                out[f"ghosts_idx_{r}"] = cat(g_idx)
                out[f"ghosts_size_{r}"] = np.asarray(g_size, dtype=np.int64)
                out[f"unique_ghosts_{r}"] = np.asarray(g_ranks, dtype=np.int64)
            tag = f"halo_plan_P{P}_{shape[0]}x{shape[1]}x{shape[2]}_grid{grid[0]}x{grid[1]}x{grid[2]}_{ghost_order}"
            np.savez_compressed(os.path.join(HERE, tag + ".npz"), **out)
            print("wrote", tag)


    # ---- time-loop fixtures: the reference's OWN operators and scatter closures driven through the stage sequence
    # of its RK4 loop (cuda/demo_linear_box.py:487-566 == numba-cpu/demo_linear_box.py's f0 / f1 / rk4; the demos
    # themselves need dolfinx for the mesh and cannot run here).  Every array operation below is a call into the imported
    # reference (copy / axpy / fill / pointwise_divide / mass_operator / stiffness_operator / scatter_forward /
    # scatter_reverse); what this script contributes is the ORDER of the calls, written next to the reference's lines,
    # and the inputs.  It pins the operator-composition half of the time loop (which vector goes where, t vs tn, signs of
    # the facet terms as the demo passes them); the material constants and the source formula are restated from
    # cuda/demo_linear_box.py:336-345,512-533.
    if only in ("all", "rk4"):
        ls = fusgpu_loader.submodule("linear_solver")
        c0, rho0, f0, p0 = 1500.0, 1000.0, 0.5e6, 60000.0
        w0 = 2 * np.pi * f0
        rk_cases = [("P2_2x2x2_pert_1rank", 2, (2, 2, 2), (1, 1, 1), 0.12, 12), ("P2_4x2x2_pert_2ranks", 2, (4, 2, 2), (2, 1, 1), 0.12, 8)]
        for tag, P, shape, grid, perturb, nsteps in rk_cases:
            n = P + 1
            R = int(np.prod(grid))
            L = 0.003 * shape[0]
            lengths = tuple(L * s_ / shape[0] for s_ in shape)
            meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r, length=lengths, perturb=perturb, seed=11) for r in range(R)]
            od_all, gd_all = utils.compute_scatterer_data_all([m.index_map for m in meshes])
            pts, wts, D = gll.tabulate_1d(P)
            wts3, wts2 = gll.tensor_weights_3d(wts), gll.tensor_weights_2d(wts)
            dphi_g, dphi_f = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts)), pre.tabulate_facet_gradients(pts)
            h = min(ls.time_step_parameters(m, P, c0, f0, L) for m in meshes)
            dt, _, _ = ls.snap_time_step(h, P, c0, f0, L)
            rk = []  # per rank: the arrays the demo builds at set-up (cuda/demo_linear_box.py:245-345)
            for m in meshes:
                nc = m.ncells
                G, detJ = np.zeros((nc, n**3, 6)), np.zeros((nc, n**3))
                ref_pre.compute_scaled_geometrical_factor(G, (m.x_dofs, m.x_g), nc, dphi_g, wts3)
                ref_pre.compute_scaled_jacobian_determinant(detJ, (m.x_dofs, m.x_g), nc, dphi_g, wts3)
                bd1, bd2 = m.boundary_facets([2]), m.boundary_facets([3])  # x = 0 source, x = L absorbing
                dF1, dF2 = np.zeros((bd1.shape[0], n * n)), np.zeros((bd2.shape[0], n * n))
                if bd1.shape[0]:
                    ref_pre.compute_boundary_facets_scaled_jacobian_determinant(dF1, (m.x_dofs, m.x_g), bd1, dphi_f, wts2)
                if bd2.shape[0]:
                    ref_pre.compute_boundary_facets_scaled_jacobian_determinant(dF2, (m.x_dofs, m.x_g), bd2, dphi_f, wts2)
                rk.append(dict(m=m, G=G, detJ=detJ, dF1=dF1, dF2=dF2, fd1=m.facet_dofmap(bd1), fd2=m.facet_dofmap(bd2),
                               cc1=np.full(nc, 1.0 / rho0 / c0 / c0), cc2=np.full(nc, -1.0 / rho0),      # :336-337
                               fc1=np.full(bd1.shape[0], 1.0 / rho0), fc2=np.full(bd2.shape[0], -1.0 / rho0 / c0)))  # :339-345
            world = FakeWorld()
            fwd = [ref_sc.scatter_forward(FakeComm(world, r), od_all[r], gd_all[r], meshes[r].nlocal, np.float64) for r in range(R)]
            rev = [ref_sc.scatter_reverse(FakeComm(world, r), od_all[r], gd_all[r], meshes[r].nlocal, np.float64) for r in range(R)]

            def scatter_all(closures, arrays):
                """every rank's reference closure on its vector; the closure packs, sends, receives and unpacks in one
                call, so a first pass over scratch copies puts every rank's sends into the mailbox"""
                if R == 1:
                    return
                world.box.clear()
                scratch = [a.copy() for a in arrays]
                for r in range(R):
                    try:
                        closures[r](scratch[r])
                    except KeyError:
                        pass
                for r in range(R):
                    closures[r](arrays[r])

            stiff = ref_ops.stiffness_operator(P, D.flatten(), np.float64)
            mass_c, mass_f = ref_ops.mass_operator(n**3, np.float64), ref_ops.mass_operator(n * n, np.float64)
            out = {"P": P, "shape": np.array(shape), "grid": np.array(grid), "perturb": perturb, "seed": 11, "lengths": np.array(lengths),
                   "dt": dt, "nsteps": nsteps, "c0": c0, "rho0": rho0, "f0": f0, "p0": p0}
            for source_time in ("tn", "t"):
                st = []
                for d in rk:
                    nd_ = d["m"].ndofs
                    z = lambda: np.zeros(nd_)  # noqa: E731
                    st.append(dict(u=z(), v=z(), un=z(), vn=z(), u0=z(), v0=z(), ku=z(), kv=z(), g=z(), b=z(), m=z(), u_n=z(), v_n=z(),
                                   axpy=ref_ops.axpy(nd_)))
                # lumped mass (:421-428): u_t = 1; m = M(cc1) u_t; scatter_rev(m)
                for d, s_ in zip(rk, st):
                    ones = np.zeros(d["m"].ndofs)
                    ref_ops.fill(1.0, ones)
                    ref_ops.fill(0.0, s_["m"])
                    mass_c(ones, d["cc1"], s_["m"], d["detJ"], d["m"].dofmap)
                scatter_all(rev, [s_["m"] for s_ in st])
                a_runge, b_runge, c_runge = (0.0, 0.5, 0.5, 1.0), (1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0), (0.0, 0.5, 0.5, 1.0)
                t = 0.0
                for _ in range(nsteps):
                    for s_ in st:
                        ref_ops.copy(s_["u"], s_["u0"])  # :491-492
                        ref_ops.copy(s_["v"], s_["v0"])
                    for i in range(4):
                        tn = t + c_runge[i] * dt
                        ts_ = tn if source_time == "tn" else t  # the CUDA demo evaluates window and g at t (:515-532), numba-cpu / C++ at tn
                        T_, alpha = 1.0 / f0, 4.0
                        window = 0.5 * (1.0 - np.cos(f0 * np.pi * ts_ / alpha)) if ts_ < T_ * alpha else 1.0
                        g_vals = window * p0 * w0 / c0 * np.cos(w0 * ts_)
                        for d, s_ in zip(rk, st):
                            ref_ops.copy(s_["u0"], s_["un"])  # :496-500
                            ref_ops.copy(s_["v0"], s_["vn"])
                            s_["axpy"](a_runge[i] * dt, s_["ku"], s_["un"])
                            s_["axpy"](a_runge[i] * dt, s_["kv"], s_["vn"])
                            ref_ops.copy(s_["vn"], s_["ku"])  # f0, :508
                            ref_ops.fill(g_vals, s_["g"])  # :533
                            ref_ops.copy(s_["un"], s_["u_n"])  # :536-537
                            ref_ops.copy(s_["vn"], s_["v_n"])
                        scatter_all(fwd, [s_["u_n"] for s_ in st])  # :539-540
                        scatter_all(fwd, [s_["v_n"] for s_ in st])
                        for d, s_ in zip(rk, st):
                            ref_ops.fill(0.0, s_["b"])  # :543
                            stiff(s_["u_n"], d["cc2"], s_["b"], d["G"], d["m"].dofmap)  # :545-547
                            mass_f(s_["g"], d["fc1"], s_["b"], d["dF1"], d["fd1"])  # :548-550
                            mass_f(s_["v_n"], d["fc2"], s_["b"], d["dF2"], d["fd2"])  # :551-553
                        scatter_all(rev, [s_["b"] for s_ in st])  # :555
                        for s_ in st:
                            ref_ops.pointwise_divide(s_["b"], s_["m"], s_["kv"])  # :558
                            s_["axpy"](b_runge[i] * dt, s_["ku"], s_["u"])  # :564-565
                            s_["axpy"](b_runge[i] * dt, s_["kv"], s_["v"])
                    t += dt
                scatter_all(fwd, [s_["u"] for s_ in st])
                scatter_all(fwd, [s_["v"] for s_ in st])
                for r, s_ in enumerate(st):
                    out[f"ref_u_{source_time}_{r}"] = s_["u"]
                    out[f"ref_v_{source_time}_{r}"] = s_["v"]
                    out[f"ref_m_{r}"] = s_["m"]
            np.savez_compressed(os.path.join(HERE, f"rk4_{tag}.npz"), **out)
            print("wrote", f"rk4_{tag}", "dt", dt, "max|u|", max(float(np.abs(out[f"ref_u_tn_{r}"]).max()) for r in range(R)))


    # ---- the Westervelt loop the same way: cuda/demo_nonlinear_bowl.py:357-374 (coefficients), :458-475 (steady part of
    # the lumped mass, incl. the absorbing-facet term), :533-650 (stage: g and dg/dt, w_n = v_n^2, three forward scatters,
    # m = M(cc2) u_n + m0 with its own reverse scatter, two stiffness applies, cell mass of w_n, three facet masses,
    # reverse scatter of b, pointwise divide).  Every array operation is the imported reference's (numba-cpu flavour);
    # its CUDA-only ``square`` kernel (cuda/operators.py) is the one pointwise op done in numpy (w_n = v_n * v_n).
    if only in ("all", "rk4nl"):
        ls = fusgpu_loader.submodule("linear_solver")
        c0, rho0, f0, beta0, att_dB = 1480.0, 1000.0, 1.1e6, 3.5, 0.2
        p0 = rho0 * c0 * 0.38557513826589934  # :56-62
        w0 = 2 * np.pi * f0
        delta0 = 2 * (att_dB / 20 * np.log(10)) * c0**3 / w0 / w0  # compute_diffusivity_of_sound, cuda/utils.py:157-162
        for tag, P, shape, grid, nsteps in (("P2_2x2x2_bowl_1rank", 2, (2, 2, 2), (1, 1, 1), 10), ("P2_4x2x2_bowl_2ranks", 2, (4, 2, 2), (2, 1, 1), 8),
                                            ("P3_4x2x2_bowl_2ranks", 3, (4, 2, 2), (2, 1, 1), 6)):  # P = 3: the solvers' default cell kernel forms G itself
            if match and match not in f"rk4nl_{tag}":
                continue
            n = P + 1
            R = int(np.prod(grid))
            L = 0.0015 * shape[0]
            lengths = tuple(L * s_ / shape[0] for s_ in shape)
            amp = 0.15 * (L / shape[0])

            def bowl(xg, L=L, lengths=lengths, amp=amp):
                out = xg.copy()
                yy, zz = xg[:, 1] / lengths[1] - 0.5, xg[:, 2] / lengths[2] - 0.5
                out[:, 0] = xg[:, 0] + amp * 4 * (yy * yy + zz * zz) * (1.0 - xg[:, 0] / L)
                return out

            meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r, length=lengths, warp=bowl) for r in range(R)]
            od_all, gd_all = utils.compute_scatterer_data_all([m.index_map for m in meshes])
            pts, wts, D = gll.tabulate_1d(P)
            wts3, wts2 = gll.tensor_weights_3d(wts), gll.tensor_weights_2d(wts)
            dphi_g, dphi_f = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts)), pre.tabulate_facet_gradients(pts)
            h = min(ls.time_step_parameters(m, P, c0, f0, L) for m in meshes)
            dt = 0.40 * h / (c0 * P**2)  # :119-125
            dt = (1.0 / f0) / (int((1.0 / f0) / dt) + 1)
            rk = []
            for m in meshes:
                nc = m.ncells
                G, detJ = np.zeros((nc, n**3, 6)), np.zeros((nc, n**3))
                ref_pre.compute_scaled_geometrical_factor(G, (m.x_dofs, m.x_g), nc, dphi_g, wts3)
                ref_pre.compute_scaled_jacobian_determinant(detJ, (m.x_dofs, m.x_g), nc, dphi_g, wts3)
                bd1, bd2 = m.boundary_facets([2]), m.boundary_facets([3])
                dF1, dF2 = np.zeros((bd1.shape[0], n * n)), np.zeros((bd2.shape[0], n * n))
                if bd1.shape[0]:
                    ref_pre.compute_boundary_facets_scaled_jacobian_determinant(dF1, (m.x_dofs, m.x_g), bd1, dphi_f, wts2)
                if bd2.shape[0]:
                    ref_pre.compute_boundary_facets_scaled_jacobian_determinant(dF2, (m.x_dofs, m.x_g), bd2, dphi_f, wts2)
                full = lambda k, v: np.full(k, v)  # noqa: E731
                rk.append(dict(m=m, G=G, detJ=detJ, dF1=dF1, dF2=dF2, fd1=m.facet_dofmap(bd1), fd2=m.facet_dofmap(bd2),
                               cc1=full(nc, 1.0 / rho0 / c0 / c0), cc2=full(nc, -2.0 * beta0 / rho0 / rho0 / c0**4),          # :358-359
                               cc3=full(nc, -1.0 / rho0), cc4=full(nc, -delta0 / rho0 / c0 / c0),                              # :360-361
                               cc5=full(nc, 2.0 * beta0 / rho0 / rho0 / c0**4),                                                # :362
                               fc1_1=full(bd1.shape[0], 1.0 / rho0), fc2_1=full(bd1.shape[0], delta0 / rho0 / c0 / c0),        # :366-368
                               fc1_2=full(bd2.shape[0], delta0 / rho0 / c0**3), fc2_2=full(bd2.shape[0], -1.0 / rho0 / c0)))   # :372-374
            world = FakeWorld()
            fwd = [ref_sc.scatter_forward(FakeComm(world, r), od_all[r], gd_all[r], meshes[r].nlocal, np.float64) for r in range(R)]
            rev = [ref_sc.scatter_reverse(FakeComm(world, r), od_all[r], gd_all[r], meshes[r].nlocal, np.float64) for r in range(R)]

            def scatter_all(closures, arrays):
                if R == 1:
                    return
                world.box.clear()
                scratch = [a.copy() for a in arrays]
                for r in range(R):
                    try:
                        closures[r](scratch[r])
                    except KeyError:
                        pass
                for r in range(R):
                    closures[r](arrays[r])

            stiff = ref_ops.stiffness_operator(P, D.flatten(), np.float64)
            mass_c, mass_f = ref_ops.mass_operator(n**3, np.float64), ref_ops.mass_operator(n * n, np.float64)
            st = []
            for d in rk:
                nd_ = d["m"].ndofs
                z = lambda nd_=nd_: np.zeros(nd_)  # noqa: E731
                st.append(dict(u=z(), v=z(), un=z(), vn=z(), u0=z(), v0=z(), ku=z(), kv=z(), g=z(), dg=z(), b=z(), m=z(), m0=z(), u_n=z(),
                               v_n=z(), w_n=z(), axpy=ref_ops.axpy(nd_)))
            for d, s_ in zip(rk, st):  # :458-468  m0 = M(cc1) 1 + M_f2(fc1_2) 1, scatter_rev
                ones = np.zeros(d["m"].ndofs)
                ref_ops.fill(1.0, ones)
                mass_c(ones, d["cc1"], s_["m0"], d["detJ"], d["m"].dofmap)
                if d["fd2"].size:
                    mass_f(ones, d["fc1_2"], s_["m0"], d["dF2"], d["fd2"])
            scatter_all(rev, [s_["m0"] for s_ in st])
            a_runge, b_runge, c_runge = (0.0, 0.5, 0.5, 1.0), (1.0 / 6.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 6.0), (0.0, 0.5, 0.5, 1.0)
            t = 0.0
            for _ in range(nsteps):
                for s_ in st:
                    ref_ops.copy(s_["u"], s_["u0"])
                    ref_ops.copy(s_["v"], s_["v0"])
                for i in range(4):
                    tn = t + c_runge[i] * dt  # this package's default (numba-cpu / C++ convention); the CUDA demo uses t
                    T_, alpha = 1.0 / f0, 4.0
                    if tn < T_ * alpha:
                        window = 0.5 * (1.0 - np.cos(f0 * np.pi * tn / alpha))
                        dwindow = 0.5 * np.pi * f0 / alpha * np.sin(f0 * np.pi * tn / alpha)
                    else:
                        window, dwindow = 1.0, 0.0
                    g_vals = window * 2.0 * p0 * w0 / c0 * np.cos(w0 * tn)  # :569-576
                    dg_vals = dwindow * 2.0 * p0 * w0 / c0 * np.cos(w0 * tn) - window * 2.0 * p0 * w0**2 / c0 * np.sin(w0 * tn)  # :577-591
                    for d, s_ in zip(rk, st):
                        ref_ops.copy(s_["u0"], s_["un"])
                        ref_ops.copy(s_["v0"], s_["vn"])
                        s_["axpy"](a_runge[i] * dt, s_["ku"], s_["un"])
                        s_["axpy"](a_runge[i] * dt, s_["kv"], s_["vn"])
                        ref_ops.copy(s_["vn"], s_["ku"])
                        ref_ops.fill(g_vals, s_["g"])
                        ref_ops.fill(dg_vals, s_["dg"])
                        ref_ops.copy(s_["un"], s_["u_n"])  # :597-599
                        ref_ops.copy(s_["vn"], s_["v_n"])
                        s_["w_n"][:] = s_["vn"] * s_["vn"]  # square[...](vn_d, w_n_d): CUDA-only kernel, done in numpy
                    for key in ("u_n", "v_n", "w_n"):
                        scatter_all(fwd, [s_[key] for s_ in st])  # :600-602
                    for d, s_ in zip(rk, st):
                        ref_ops.fill(0.0, s_["m"])  # :605-608
                        mass_c(s_["u_n"], d["cc2"], s_["m"], d["detJ"], d["m"].dofmap)
                    scatter_all(rev, [s_["m"] for s_ in st])  # :609
                    for d, s_ in zip(rk, st):
                        s_["axpy"](1.0, s_["m0"], s_["m"])  # :611
                        ref_ops.fill(0.0, s_["b"])  # :614
                        stiff(s_["u_n"], d["cc3"], s_["b"], d["G"], d["m"].dofmap)  # :616-621
                        stiff(s_["v_n"], d["cc4"], s_["b"], d["G"], d["m"].dofmap)
                        mass_c(s_["w_n"], d["cc5"], s_["b"], d["detJ"], d["m"].dofmap)  # :622-624
                        if d["fd1"].size:
                            mass_f(s_["g"], d["fc1_1"], s_["b"], d["dF1"], d["fd1"])  # :625-631
                            mass_f(s_["dg"], d["fc2_1"], s_["b"], d["dF1"], d["fd1"])
                        if d["fd2"].size:
                            mass_f(s_["v_n"], d["fc2_2"], s_["b"], d["dF2"], d["fd2"])  # :632-635
                    scatter_all(rev, [s_["b"] for s_ in st])  # :636
                    for s_ in st:
                        ref_ops.pointwise_divide(s_["b"], s_["m"], s_["kv"])  # :639
                        s_["axpy"](b_runge[i] * dt, s_["ku"], s_["u"])  # :645-646
                        s_["axpy"](b_runge[i] * dt, s_["kv"], s_["v"])
                t += dt
            scatter_all(fwd, [s_["u"] for s_ in st])
            scatter_all(fwd, [s_["v"] for s_ in st])
            out = {"P": P, "shape": np.array(shape), "grid": np.array(grid), "lengths": np.array(lengths), "bowl_amplitude": amp, "dt": dt,
                   "nsteps": nsteps, "c0": c0, "rho0": rho0, "f0": f0, "p0": p0, "beta": beta0, "att_dB": att_dB}
            for r, s_ in enumerate(st):
                out[f"ref_u_tn_{r}"], out[f"ref_v_tn_{r}"], out[f"ref_m0_{r}"] = s_["u"], s_["v"], s_["m0"]
            np.savez_compressed(os.path.join(HERE, f"rk4nl_{tag}.npz"), **out)
            print("wrote", f"rk4nl_{tag}", "dt", dt, "max|u|", max(float(np.abs(out[f"ref_u_tn_{r}"]).max()) for r in range(R)))


if __name__ == "__main__":
    main()
