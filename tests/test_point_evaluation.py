"""Point evaluation (the drivers' output path): counterpart of cuda/utils.py:117-154 ``compute_eval_params`` and of
``Function.eval`` (cuda/demo_linear_box.py:128-141, 587-605), without dolfinx.  Known-answer tests: a degree-P
polynomial in the REFERENCE coordinates is reproduced exactly by the degree-P GLL interpolant, so on an affine mesh any
polynomial of degree <= P in x, y, z is evaluated exactly at arbitrary points; on a perturbed (trilinear) mesh the cell
location / Newton inversion is checked by evaluating the coordinate field itself (x(xi) is in the space)."""
import numpy as np
import pytest

from conftest import pkg


@pytest.mark.parametrize("P", [2, 4])
def test_polynomial_is_evaluated_exactly_on_an_affine_mesh(P):
    boxmesh, pe = pkg("boxmesh"), pkg("point_evaluation")
    mesh = boxmesh.BoxMesh(P, (3, 4, 2), length=(0.3, 0.4, 0.2))
    xyz = mesh.dof_coordinates()
    f = lambda p: 1.0 + 2 * p[:, 0] ** P - 3 * p[:, 0] * p[:, 1] ** (P - 1) + p[:, 2] ** 2 * p[:, 1] ** (P - 2)  # noqa: E731
    u = f(xyz)
    rng = np.random.default_rng(0)
    pts = rng.random((3, 500)) * np.array([[0.3], [0.4], [0.2]])
    pts[:, :8] = np.array([[0, 0.3, 0, 0.3, 0.1, 0.2, 0.3, 0.15], [0, 0, 0.4, 0.4, 0.1, 0.2, 0.4, 0.2], [0, 0, 0, 0.2, 0.1, 0.1, 0.2, 0.0]])  # corners, cell faces
    x_eval, cells = pe.compute_eval_params(mesh, pts, np.float64)
    assert x_eval.shape == (500, 3) and len(cells) == 500  # every point of the box is found
    vals = pe.eval_function(mesh, u, x_eval, cells)
    assert np.max(np.abs(vals - f(x_eval))) < 1e-12


def test_points_outside_the_rank_are_dropped_and_partitions_cover_the_box():
    """cuda/utils.py:146-151: only the points that collide with a cell of this process are kept; over the ranks of a
    partition every point is kept at least once and evaluates to the same value."""
    boxmesh, pe = pkg("boxmesh"), pkg("point_evaluation")
    P, cells, grid = 3, (4, 4, 2), (2, 2, 1)
    serial = boxmesh.BoxMesh(P, cells, perturb=0.12, seed=5)
    f = lambda p: np.sin(3 * p[:, 0]) * np.cos(2 * p[:, 1]) + p[:, 2]  # noqa: E731
    rng = np.random.default_rng(1)
    pts = 0.06 + 0.88 * rng.random((3, 300))  # BoxMesh perturbs its boundary vertices too: stay inside the perturbed domain
    xs, cs = pe.compute_eval_params(serial, pts)
    assert len(cs) == 300
    ref = pe.eval_function(serial, f(serial.dof_coordinates()), xs, cs)
    assert np.max(np.abs(ref - f(xs))) < 5e-3  # interpolation error of a smooth field at P = 3, h = 1/4
    seen = np.zeros(300, dtype=int)
    for r in range(4):
        m = boxmesh.BoxMesh(P, cells, grid=grid, rank=r, perturb=0.12, seed=5)
        xr, cr = pe.compute_eval_params(m, pts)
        assert 0 < len(cr) < 300
        vr = pe.eval_function(m, f(m.dof_coordinates()), xr, cr)
        # match the kept points back to the global list
        idx = np.array([int(np.argmin(np.linalg.norm(pts.T - p, axis=1))) for p in xr])
        seen[idx] += 1
        assert np.max(np.abs(vr - ref[idx])) < 1e-12  # the same interpolant, whichever rank evaluates it
    assert np.all(seen >= 1)
    outside = np.array([[1.5], [0.5], [0.5]])
    assert pe.compute_eval_params(serial, outside)[1] == []


def test_coordinate_field_on_a_perturbed_mesh():
    """Cell location + Newton inversion on non-affine cells: the geometry map itself is in the space (trilinear <= P),
    so evaluating the dof-coordinate field at a point returns the point."""
    boxmesh, pe = pkg("boxmesh"), pkg("point_evaluation")
    mesh = boxmesh.BoxMesh(2, 5, perturb=0.16, seed=3)
    xyz = mesh.dof_coordinates()
    rng = np.random.default_rng(2)
    pts = 0.06 + 0.88 * rng.random((3, 400))
    xs, cs = pe.compute_eval_params(mesh, pts)
    assert len(cs) == 400
    for a in range(3):
        assert np.max(np.abs(pe.eval_function(mesh, xyz[:, a], xs, cs) - xs[:, a])) < 1e-12


def test_the_demo_plane_sampling():
    """cuda/demo_linear_box.py:128-141: 100 x 100 points on the z = 0 plane of the box, rows (x, y, value)."""
    boxmesh, pe = pkg("boxmesh"), pkg("point_evaluation")
    L = 0.12
    mesh = boxmesh.BoxMesh(3, 6, length=L)
    xp = np.linspace(0, L, 100)
    X, Y = np.meshgrid(xp, xp)
    points = np.zeros((3, 100 * 100))
    points[0], points[1] = X.flatten(), Y.flatten()
    x_eval, cell_eval = pe.compute_eval_params(mesh, points, np.float64)
    assert x_eval.shape == (10000, 3)
    u = mesh.dof_coordinates()[:, 0] ** 2
    data = np.zeros_like(x_eval)
    data[:, 0], data[:, 1] = x_eval[:, 0], x_eval[:, 1]
    data[:, 2] = pe.eval_function(mesh, u, x_eval, cell_eval)
    assert np.max(np.abs(data[:, 2] - data[:, 0] ** 2)) < 1e-13
