"""Host precompute (G, detJ, facet detJ) vs the reference's outputs (golden) and
vs closed forms on affine cells."""

import numpy as np
import pytest

from conftest import golden_files, pkg, rel_l2


@pytest.mark.parametrize("path", golden_files("ops_"), ids=lambda p: p.split("/")[-1][:-4])
def test_precompute_vs_reference(path):
    pre = pkg("precompute")
    d = np.load(path)
    dt = d["x"].dtype
    tol = 2e-15 if dt == np.float64 else 2e-6
    nc = d["dofmap"].shape[0]
    mesh = (d["x_dofs"], d["x_g"])
    detJ, G, dF = np.zeros_like(d["ref_detJ"]), np.zeros_like(d["ref_G"]), np.zeros_like(d["ref_detJ_f"])
    pre.compute_scaled_jacobian_determinant(detJ, mesh, nc, d["dphi_geom"], d["wts3"])
    pre.compute_scaled_geometrical_factor(G, mesh, nc, d["dphi_geom"], d["wts3"])
    pre.compute_boundary_facets_scaled_jacobian_determinant(dF, mesh, d["boundary_data"], d["dphi_facet"], d["wts2"])
    assert rel_l2(detJ, d["ref_detJ"]) < tol
    assert rel_l2(G, d["ref_G"]) < tol
    assert rel_l2(dF, d["ref_detJ_f"]) < tol


def test_affine_cell_closed_form():
    """h = (.5, .25, .2): G = diag(hx hy hz / h_a^2) w, detJ = hx hy hz w."""
    gll, boxmesh, pre = pkg("gll"), pkg("boxmesh"), pkg("precompute")
    P = 3
    mesh = boxmesh.BoxMesh(P, (2, 4, 5), length=(1.0, 1.0, 1.0))
    pts, wts, _ = gll.tabulate_1d(P)
    w3 = gll.tensor_weights_3d(wts)
    dg = pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))
    n3 = (P + 1) ** 3
    G, detJ = np.zeros((mesh.ncells, n3, 6)), np.zeros((mesh.ncells, n3))
    pre.compute_scaled_geometrical_factor(G, (mesh.x_dofs, mesh.x_g), mesh.ncells, dg, w3)
    pre.compute_scaled_jacobian_determinant(detJ, (mesh.x_dofs, mesh.x_g), mesh.ncells, dg, w3)
    h = np.array([0.5, 0.25, 0.2])
    vol = h.prod()
    assert np.allclose(detJ, vol * w3[None, :], rtol=1e-13)
    assert abs(detJ.sum() - 1.0) < 1e-13
    assert np.allclose(G[..., 0], vol / h[0] ** 2 * w3, rtol=1e-13)
    assert np.allclose(G[..., 3], vol / h[1] ** 2 * w3, rtol=1e-13)
    assert np.allclose(G[..., 5], vol / h[2] ** 2 * w3, rtol=1e-13)
    assert np.max(np.abs(G[..., [1, 2, 4]])) < 1e-15
    # boundary facets: total area of the unit box = 6
    bd = mesh.boundary_facets()
    dF = np.zeros((bd.shape[0], (P + 1) ** 2))
    pre.compute_boundary_facets_scaled_jacobian_determinant(
        dF, (mesh.x_dofs, mesh.x_g), bd, pre.tabulate_facet_gradients(pts), gll.tensor_weights_2d(wts))
    assert abs(dF.sum() - 6.0) < 1e-12


def test_geometry_tabulation_partition_of_unity():
    pre = pkg("precompute")
    X = np.random.default_rng(0).random((20, 3))
    d = pre.tabulate_hex_p1_gradients(X)
    assert d.shape == (3, 20, 8)
    assert np.max(np.abs(d.sum(axis=2))) < 1e-14  # gradients of a partition of unity sum to 0
