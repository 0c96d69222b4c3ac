"""Pin the ORACLE (numpy and C restatements) against the golden vectors that were
produced by running the reference itself (tests/golden/generate_golden.py), and
its two primitives against the reference's own C++ templates where
oracle/_ref/libref_sumfact.so could be built."""

import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT, golden_files, rel_l2
from oracle import oracle_np

TOL = {np.dtype(np.float64): 2e-15, np.dtype(np.float32): 1e-6}


@pytest.mark.parametrize("path", golden_files("ops_"), ids=lambda p: p.split("/")[-1][:-4])
def test_operators_vs_reference_outputs(path, oracle_c):
    d = np.load(path)
    P, dt = int(d["P"]), d["x"].dtype
    n, tol = P + 1, TOL[d["x"].dtype]
    G = np.ascontiguousarray(d["ref_G"])
    for impl in ("np", "c", "c_omp"):
        y = d["y0"].copy()
        if impl == "np":
            oracle_np.stiffness_apply(P, d["dphi_1d"].flatten(), d["x"], d["cell_constants"], y, G, d["dofmap"])
        else:
            oracle_c.stiffness_apply(P, d["dphi_1d"], d["x"], d["cell_constants"], y, G, d["dofmap"], threads=1 if impl == "c" else 3)
        assert rel_l2(y, d["ref_y_stiffness"]) < tol, impl
        y = d["y0"].copy()
        (oracle_np if impl == "np" else oracle_c).mass_apply(d["x"], d["cell_constants"], y, d["ref_detJ"], d["dofmap"])
        assert rel_l2(y, d["ref_y_mass"]) < tol, impl
        y = d["y0"].copy()
        (oracle_np if impl == "np" else oracle_c).mass_apply(d["x"], d["facet_constants"], y, d["ref_detJ_f"], d["bfacet_dofmap"])
        assert rel_l2(y, d["ref_y_facet_mass"]) < tol, impl
    y = d["vb"].copy()
    oracle_c.axpy(float(d["alpha"]), d["va"], y)
    assert rel_l2(y, d["ref_y_axpy"]) < tol
    out = np.zeros_like(d["va"])
    oracle_c.pointwise_divide(d["va"], d["vb"], out)
    assert rel_l2(out, d["ref_y_divide"]) < tol


@pytest.mark.parametrize("path", golden_files("scatter_"), ids=lambda p: p.split("/")[-1][:-4])
def test_scatter_vs_reference_closures(path):
    """Simulated-rank scatter_reverse / scatter_forward against what the reference's own
    closures (numba-cpu/scatterer.py:78-207) produced on the same partition."""
    from conftest import pkg

    d = np.load(path)
    P, shape, grid = int(d["P"]), tuple(d["shape"]), tuple(d["grid"])
    boxmesh, utils = pkg("boxmesh"), pkg("utils")
    R = int(np.prod(grid))
    meshes = [boxmesh.BoxMesh(P, shape, grid=grid, rank=r) for r in range(R)]
    od, gd = utils.compute_scatterer_data_all([m.index_map for m in meshes])
    nl = [m.nlocal for m in meshes]
    rev = [d[f"in_{r}"].copy() for r in range(R)]
    fwd = [d[f"in_{r}"].copy() for r in range(R)]
    oracle_np.scatter_reverse_all(rev, od, gd, nl)
    oracle_np.scatter_forward_all(fwd, od, gd, nl)
    for r in range(R):
        assert int(d[f"nlocal_{r}"]) == nl[r]
        assert np.allclose(rev[r], d[f"ref_rev_{r}"], rtol=0, atol=1e-14)
        assert np.array_equal(fwd[r], d[f"ref_fwd_{r}"])


def test_primitives_vs_reference_cpp(oracle_c):
    """contract / transpose of the C oracle vs the reference's C++ templates
    (cpp/common/sum_factorisation.hpp:43-49,70-86) compiled into oracle/_ref."""
    so = os.path.join(ROOT, "oracle", "_ref", "libref_sumfact.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (reference checkout absent on this machine)")
    ref = C.CDLL(so)
    vp = C.c_void_p
    ref.ref_contract_f64.argtypes = [C.c_int, C.c_int, vp, vp, vp]
    ref.ref_transpose_f64.argtypes = [C.c_int, C.c_int, vp, vp]
    rng = np.random.default_rng(0)
    for n in (3, 4, 5, 7):
        A, B = rng.standard_normal(n * n), rng.standard_normal(n**3)
        for tr in (0, 1):
            c_ref, c_mine = rng.standard_normal(n**3), None
            c_mine = c_ref.copy()
            assert ref.ref_contract_f64(n, tr, A.ctypes.data, B.ctypes.data, c_ref.ctypes.data) == 0
            oracle_c.lib.oracle_contract_f64(n, n, n, n, tr, A.ctypes.data, B.ctypes.data, c_mine.ctypes.data)
            assert np.allclose(c_mine, c_ref, rtol=1e-14, atol=1e-14)
        for which, offs in ((0, (n, n * n, 1)), (1, (1, n, n * n))):
            t_ref, t_mine = np.zeros(n**3), np.zeros(n**3)
            assert ref.ref_transpose_f64(n, which, B.ctypes.data, t_ref.ctypes.data) == 0
            oracle_c.lib.oracle_transpose_f64(n, n, n, *offs, B.ctypes.data, t_mine.ctypes.data)
            assert np.array_equal(t_mine, t_ref)


def test_oracle_pack_unpack(oracle_c):
    rng = np.random.default_rng(1)
    buf = rng.standard_normal(50)
    idx = rng.integers(0, 50, size=20)
    out = np.zeros(20)
    oracle_c.pack(buf, out, idx)
    assert np.array_equal(out, buf[idx])
    a, b = buf.copy(), buf.copy()
    oracle_c.unpack_rev(out, a, idx)
    np.add.at(b, idx, out)
    assert np.allclose(a, b, atol=1e-15)
