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

    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else "all"  # all | ops | scatter | plan

    # ---- operator + precompute fixtures --------------------------------------
    cases = []
    for P in (2, 3, 4, 6):
        for shape in ((2, 2, 2), (3, 2, 2)):
            for perturb in (0.0, 0.16):
                for dt in (np.float64, np.float32):
                    if P == 6 and shape == (3, 2, 2):
                        continue  # keep the fixture set small
                    cases.append((P, shape, perturb, dt))
    for P, shape, perturb, dt in (cases if only in ("all", "ops") else []):
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


if __name__ == "__main__":
    main()
