import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import fusgpu_loader  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native pieces are normally built by __graft_entry__.build(); build them here if a fresh
    # checkout runs the tests first (hipcc cross-compiles gfx950 without a GPU)
    import subprocess

    import __graft_entry__ as entry

    # a prebuilt libfusgpu.so travels with the tree (git-ignored, not gpurun-ignored): rebuild it unless it was built
    # from exactly these sources (fus_source_hash() vs the tree's hash), here and on the GPU box alike
    entry.ensure_library(log=lambda m: print(f"[conftest] {m}", file=sys.stderr))
    orc = os.path.join(ROOT, "oracle", "_build", "libfus_oracle.so")
    if not os.path.exists(orc):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_build/libfus_oracle.so"], check=False, capture_output=True)
    if os.path.isdir("/root/reference") and not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_sumfact.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], check=False, capture_output=True)


def pkg(name):
    return fusgpu_loader.submodule(name)


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def rel_max(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


# Parity tolerances (SURVEY 8d / BASELINE.md 2): summation order differs from the
# reference (atomics, FMA contraction, fastmath), so the bar is relative, not bitwise.
TOL = {
    np.dtype(np.float64): dict(l2=1e-12, mx=1e-11),
    np.dtype(np.float32): dict(l2=1e-5, mx=1e-4),
}


@pytest.fixture(scope="session")
def oracle_c():
    from oracle.oracle_c import OracleLib

    return OracleLib()


def ref_field(xyz):
    """The reference's test field, numba-cpu/test_operators.py:274-279."""
    return 100 * np.sin(2 * np.pi * xyz[:, 0]) * np.cos(3 * np.pi * xyz[:, 1]) * np.sin(4 * np.pi * xyz[:, 2])


def build_problem(P, ncells, dtype=np.float64, perturb=0.0, seed=0, grid=(1, 1, 1), rank=0, random_constants=True):
    """Synthetic inputs for one rank: mesh, tables, G, detJ, x, constants (host numpy)."""
    gll, boxmesh, pre = pkg("gll"), pkg("boxmesh"), pkg("precompute")
    mesh = boxmesh.BoxMesh(P, ncells, grid=grid, rank=rank, perturb=perturb, seed=seed, dtype=dtype)
    pts, wts, D = gll.tabulate_1d(P, dtype)
    n = P + 1
    wts3 = gll.tensor_weights_3d(wts).astype(dtype)
    dphi_g = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts), dtype)
    G = np.zeros((mesh.ncells, n**3, 6), dtype=dtype)
    detJ = np.zeros((mesh.ncells, n**3), dtype=dtype)
    pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), mesh.ncells, dphi_g, wts3)
    pre.compute_scaled_jacobian_determinant(detJ, (mesh.x_dofs, mesh.x_g), mesh.ncells, dphi_g, wts3)
    x = ref_field(mesh.dof_coordinates()).astype(dtype)
    rng = np.random.default_rng(1234)
    cc = (1.0 + 0.25 * rng.standard_normal(mesh.ncells)).astype(dtype) if random_constants else np.ones(mesh.ncells, dtype)
    return dict(mesh=mesh, pts=pts, wts=wts, D=D, G=G, detJ=detJ, x=x, cc=cc, P=P, n=n)
