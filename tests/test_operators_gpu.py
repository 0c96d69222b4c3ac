"""GPU parity tests: HIP kernels (through the C ABI) vs the oracle and vs the
golden vectors produced by the reference itself.  Design follows the reference's
own tests (cuda/test_operators.py:213-356: mass, stiffness, boundary-facet mass,
rel-l2 against a trusted assembly) with the dolfinx assembly replaced by the
oracle / golden data."""

import numpy as np
import pytest

from conftest import TOL, build_problem, golden_files, pkg, ref_field, rel_l2, rel_max

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[True, False], ids=["plan", "noplan"])
def plan_mode(request):
    """Run a test on the planned (default) and the plan-free stiffness path."""
    ops = pkg("operators")
    ops.use_plan(request.param)
    yield request.param
    ops.use_plan(True)


@pytest.fixture(scope="module")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (no CPU fallback exists)")
    torch.cuda.set_device(0)
    return pkg("device"), pkg("operators")


def _check(got, ref, dtype, what):
    tol = TOL[np.dtype(dtype)]
    e2, em = rel_l2(got, ref), rel_max(got, ref)
    assert e2 < tol["l2"] and em < tol["mx"], f"{what}: rel l2 {e2:.3e} (tol {tol['l2']}), rel max {em:.3e}"


@pytest.mark.parametrize("path", golden_files("ops_"), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_reference_outputs(gpu, plan_mode, path):
    """Every golden case: same inputs the reference consumed, compare to what it produced."""
    dev, ops = gpu
    d = np.load(path)
    P, dt = int(d["P"]), d["x"].dtype
    n = P + 1
    x, dm = dev.to_device(d["x"]), dev.to_device(d["dofmap"])
    cc = dev.to_device(d["cell_constants"])
    # stiffness (numba-cpu flavour)
    y = dev.to_device(d["y0"])
    ops.stiffness_operator(P, d["dphi_1d"].flatten(), dt)(x, cc, y, dev.to_device(d["ref_G"]), dm)
    _check(y.copy_to_host(), d["ref_y_stiffness"], dt, "stiffness")
    # stiffness (cuda flavour, 2-D dphi, launch config ignored)
    y = dev.to_device(d["y0"])
    op = ops.stiffness_operator(P, dt)
    op[dm.shape[0], (n, n, n)](x, cc, y, dev.to_device(d["ref_G"]), dm, dev.to_device(d["dphi_1d"]))
    _check(y.copy_to_host(), d["ref_y_stiffness"], dt, "stiffness (cuda flavour)")
    # cell mass
    y = dev.to_device(d["y0"])
    ops.mass_operator(n**3, dt)(x, cc, y, dev.to_device(d["ref_detJ"]), dm)
    _check(y.copy_to_host(), d["ref_y_mass"], dt, "mass")
    # boundary-facet mass (cuda flavour)
    y = dev.to_device(d["y0"])
    ops.mass_operator[7, 128](x, dev.to_device(d["facet_constants"]), y, dev.to_device(d["ref_detJ_f"]),
                              dev.to_device(d["bfacet_dofmap"]))
    _check(y.copy_to_host(), d["ref_y_facet_mass"], dt, "facet mass")
    # vector ops
    va, vb = dev.to_device(d["va"]), dev.to_device(d["vb"])
    yb = dev.to_device(d["vb"])
    ops.axpy(va.numel())(float(d["alpha"]), va, yb)
    _check(yb.copy_to_host(), d["ref_y_axpy"], dt, "axpy")
    out = dev.device_array(va.shape, dt)
    ops.pointwise_divide(va, vb, out)
    _check(out.copy_to_host(), d["ref_y_divide"], dt, "pointwise_divide")


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P", list(range(1, 11)))
def test_stiffness_all_degrees_vs_oracle(gpu, oracle_c, plan_mode, P, dtype):
    """P = 1..10 on a perturbed (non-affine) mesh whose cell count is not a multiple
    of the cells-per-workgroup batch (ragged last batch)."""
    dev, ops = gpu
    ncells = (3, 2, 3) if P <= 6 else (2, 1, 2)
    pb = build_problem(P, ncells, dtype=dtype, perturb=0.16, seed=P)
    mesh = pb["mesh"]
    rng = np.random.default_rng(P)
    y0 = rng.standard_normal(mesh.ndofs).astype(dtype)
    y_ref = y0.copy()
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap)
    y = dev.to_device(y0)
    ops.stiffness_operator(P, pb["D"].flatten(), dtype)(
        dev.to_device(pb["x"]), dev.to_device(pb["cc"]), y, dev.to_device(pb["G"]), dev.to_device(mesh.dofmap))
    _check(y.copy_to_host(), y_ref, dtype, f"stiffness P={P}")


@pytest.mark.parametrize("variant,remap", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_stiffness_variants_cfg1(gpu, oracle_c, variant, remap):
    """BASELINE config 1 (P=2, 18^3 cells, ~50k dofs) for every kernel variant / XCD remap."""
    dev, ops = gpu
    lib = pkg("_lib")
    ops.use_plan(False)  # the variants are plan-free kernels
    pb = build_problem(2, 18, perturb=0.16)
    mesh = pb["mesh"]
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(2, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap, threads=4)
    old = lib.get_tuning(lib.TUNE_STIFFNESS_VARIANT), lib.get_tuning(lib.TUNE_XCD_REMAP)
    try:
        lib.set_tuning(lib.TUNE_STIFFNESS_VARIANT, variant)
        lib.set_tuning(lib.TUNE_XCD_REMAP, remap)
        y = dev.to_device(np.zeros(mesh.ndofs))
        ops.stiffness_operator(2, pb["D"].flatten(), np.float64)(
            dev.to_device(pb["x"]), dev.to_device(pb["cc"]), y, dev.to_device(pb["G"]), dev.to_device(mesh.dofmap))
        _check(y.copy_to_host(), y_ref, np.float64, f"variant {variant} remap {remap}")
    finally:
        lib.set_tuning(lib.TUNE_STIFFNESS_VARIANT, old[0])
        lib.set_tuning(lib.TUNE_XCD_REMAP, old[1])
        ops.use_plan(True)


def test_stiffness_p4_medium_vs_oracle(gpu, oracle_c, plan_mode):
    """P=4, 12^3 perturbed cells (117k dofs): full vector compare + the reference's
    invariants K.1 = 0 and symmetry v.Ku = u.Kv on the GPU result."""
    dev, ops = gpu
    pb = build_problem(4, 12, perturb=0.16)
    mesh = pb["mesh"]
    op = ops.stiffness_operator(4, pb["D"].flatten(), np.float64)
    G, dm, cc = dev.to_device(pb["G"]), dev.to_device(mesh.dofmap), dev.to_device(pb["cc"])
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(4, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap, threads=4)
    y = dev.to_device(np.zeros(mesh.ndofs))
    op(dev.to_device(pb["x"]), cc, y, G, dm)
    Ku = y.copy_to_host()
    _check(Ku, y_ref, np.float64, "stiffness P=4 12^3")
    # K * const = 0
    y1 = dev.to_device(np.zeros(mesh.ndofs))
    op(dev.to_device(np.full(mesh.ndofs, 3.0)), cc, y1, G, dm)
    assert np.max(np.abs(y1.copy_to_host())) < 1e-10 * np.max(np.abs(Ku))
    # symmetry
    v = np.random.default_rng(5).standard_normal(mesh.ndofs)
    y2 = dev.to_device(np.zeros(mesh.ndofs))
    op(dev.to_device(v), cc, y2, G, dm)
    a, b = float(v @ Ku), float(pb["x"] @ y2.copy_to_host())
    assert abs(a - b) < 1e-11 * max(abs(a), abs(b))


def test_empty_and_single_cell(gpu, oracle_c, plan_mode):
    dev, ops = gpu
    pb = build_problem(4, (1, 1, 1))
    mesh = pb["mesh"]
    op = ops.stiffness_operator(4, pb["D"].flatten(), np.float64)
    x, cc = dev.to_device(pb["x"]), dev.to_device(pb["cc"])
    G, dm = dev.to_device(pb["G"]), dev.to_device(mesh.dofmap)
    y = dev.to_device(np.zeros(mesh.ndofs))
    op(x, cc[:0], y, G[:0], dm[:0])  # zero cells: no-op
    assert np.all(y.copy_to_host() == 0)
    op(x, cc, y, G, dm)
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(4, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap)
    _check(y.copy_to_host(), y_ref, np.float64, "single cell")
    m = ops.mass_operator(125, np.float64)
    m(x, cc[:0], y, dev.to_device(pb["detJ"])[:0], dm[:0])


def test_colliding_dofmap(gpu, oracle_c, plan_mode):
    """Scatter-add under heavy collisions: every cell maps onto the same few dofs."""
    dev, ops = gpu
    P, n = 3, 4
    pb = build_problem(P, (2, 2, 2), perturb=0.1)
    rng = np.random.default_rng(3)
    ncell = 37
    dofmap = rng.integers(0, 50, size=(ncell, n**3), dtype=np.int32)
    G = np.tile(pb["G"], (5, 1, 1))[:ncell].copy()
    cc = rng.standard_normal(ncell)
    x = rng.standard_normal(50)
    y_ref = np.zeros(50)
    oracle_c.stiffness_apply(P, pb["D"], x, cc, y_ref, G, dofmap)
    y = dev.to_device(np.zeros(50))
    ops.stiffness_operator(P, pb["D"].flatten(), np.float64)(
        dev.to_device(x), dev.to_device(cc), y, dev.to_device(G), dev.to_device(dofmap))
    _check(y.copy_to_host(), y_ref, np.float64, "colliding dofmap")


def test_argument_errors(gpu):
    dev, ops = gpu
    import torch

    pb = build_problem(2, (1, 1, 1))
    mesh = pb["mesh"]
    op = ops.stiffness_operator(2, pb["D"].flatten(), np.float64)
    x, cc = dev.to_device(pb["x"]), dev.to_device(pb["cc"])
    G, dm = dev.to_device(pb["G"]), dev.to_device(mesh.dofmap)
    y = dev.to_device(np.zeros(mesh.ndofs))
    with pytest.raises(TypeError):
        op(x.float(), cc, y, G, dm)  # dtype mismatch
    with pytest.raises(TypeError):
        op(x, cc, y, G, dm.long())  # dofmap must be int32
    with pytest.raises(Exception):
        op(x.cpu(), cc, y, G, dm)  # host tensor: no CPU fallback
    with pytest.raises(ValueError):
        ops.stiffness_operator(11, np.float64)
    with pytest.raises(ValueError):
        op(x, cc, y, G[:, :5], dm)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_vector_ops(gpu, oracle_c, dtype):
    dev, ops = gpu
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 255, 256, 257, 100003):
        a = rng.standard_normal(n).astype(dtype)
        b = (2 + rng.random(n)).astype(dtype)
        for off in (0, 1):  # off=1: misaligned views take the scalar path
            ad, bd = dev.to_device(a)[off:], dev.to_device(b)[off:]
            ah, bh = a[off:].copy(), b[off:].copy()
            if ah.size == 0:
                continue
            yd = bd.clone()
            ops.axpy[1, 1](0.37, ad, yd)
            yr = bh.copy()
            oracle_c.axpy(0.37, ah, yr)
            _check(yd.cpu().numpy(), yr, dtype, "axpy")
            out = dev.device_array(ah.shape, dtype)
            ops.copy(ad, out)
            assert np.array_equal(out.copy_to_host(), ah)
            ops.fill(1.5, out)
            assert np.all(out.copy_to_host() == dtype(1.5))
            ops.pointwise_divide[1, 1](ad, bd, out)
            _check(out.copy_to_host(), ah / bh, dtype, "divide")
            ops.square[1, 1](ad, out)
            _check(out.copy_to_host(), ah * ah, dtype, "square")
            ops.scale(-2.5, ad, out)
            _check(out.copy_to_host(), dtype(-2.5) * ah, dtype, "scale")


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P", [1, 2, 4, 6, 9])
def test_mass_planned_vs_oracle(gpu, oracle_c, plan_mode, P, dtype):
    """Cell mass and boundary-facet mass on sets big enough to take the planned path
    (and the same sets plan-free), ragged last batch included."""
    dev, ops = gpu
    ncells = (13, 7, 5) if P <= 4 else (5, 3, 3)
    pb = build_problem(P, ncells, dtype=dtype, perturb=0.16, seed=P)
    mesh = pb["mesh"]
    old = ops._MASS_PLAN_MIN_ENTRIES
    ops._MASS_PLAN_MIN_ENTRIES = 1
    try:
        rng = np.random.default_rng(P)
        y0 = rng.standard_normal(mesh.ndofs).astype(dtype)
        x, cc = dev.to_device(pb["x"]), dev.to_device(pb["cc"])
        y_ref = y0.copy()
        oracle_c.mass_apply(pb["x"], pb["cc"], y_ref, pb["detJ"], mesh.dofmap)
        y = dev.to_device(y0)
        ops.mass_operator((P + 1) ** 3, dtype)(x, cc, y, dev.to_device(pb["detJ"]), dev.to_device(mesh.dofmap))
        _check(y.copy_to_host(), y_ref, dtype, f"cell mass P={P}")
        # boundary facets of the whole box
        gll, pre = pkg("gll"), pkg("precompute")
        bd = mesh.boundary_facets()
        fdm = mesh.facet_dofmap(bd)
        dF = np.zeros((bd.shape[0], (P + 1) ** 2), dtype=dtype)
        pre.compute_boundary_facets_scaled_jacobian_determinant(
            dF, (mesh.x_dofs, mesh.x_g), bd, pre.tabulate_facet_gradients(pb["pts"], dtype),
            gll.tensor_weights_2d(pb["wts"]).astype(dtype))
        fc = (1.0 + 0.3 * rng.standard_normal(bd.shape[0])).astype(dtype)
        y_ref = y0.copy()
        oracle_c.mass_apply(pb["x"], fc, y_ref, dF, fdm)
        y = dev.to_device(y0)
        ops.mass_operator((P + 1) ** 2, dtype)(x, dev.to_device(fc), y, dev.to_device(dF), dev.to_device(fdm))
        _check(y.copy_to_host(), y_ref, dtype, f"facet mass P={P}")
    finally:
        ops._MASS_PLAN_MIN_ENTRIES = old


@pytest.mark.parametrize("runs", [0, 2], ids=["rawplan", "runplan"])
@pytest.mark.parametrize("pv", [0, 1, 2])
@pytest.mark.parametrize("P", [2, 4, 6, 7])
def test_planned_kernel_builds(gpu, oracle_c, P, pv, runs):
    """Every build of the planned stiffness kernel (0 own x/y buffer, 1 LDS-aliased, 2 LDS-aliased +
    ring of G slabs) at every degree, whatever the auto dispatch would pick."""
    dev, ops = gpu
    lib = pkg("_lib")
    pb = build_problem(P, (5, 3, 4) if P < 6 else (3, 2, 3), perturb=0.16, seed=11)
    mesh = pb["mesh"]
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap)
    old = lib.get_tuning(lib.TUNE_PLAN_VARIANT)
    try:
        lib.set_tuning(lib.TUNE_PLAN_VARIANT, pv)
        lib.set_tuning(lib.TUNE_PLAN_RUNS, runs)
        ops._PLANS.clear()
        ops.use_plan(True)
        y = dev.to_device(np.zeros(mesh.ndofs))
        ops.stiffness_operator(P, pb["D"].flatten(), np.float64)(
            dev.to_device(pb["x"]), dev.to_device(pb["cc"]), y, dev.to_device(pb["G"]), dev.to_device(mesh.dofmap))
        _check(y.copy_to_host(), y_ref, np.float64, f"planned build {pv} P={P}")
        # the planned mass kernel reads the same plan
        y_ref = np.zeros(mesh.ndofs)
        oracle_c.mass_apply(pb["x"], pb["cc"], y_ref, pb["detJ"], mesh.dofmap)
        y = dev.to_device(np.zeros(mesh.ndofs))
        old_min, ops._MASS_PLAN_MIN_ENTRIES = ops._MASS_PLAN_MIN_ENTRIES, 1
        try:
            ops.mass_operator((P + 1) ** 3, np.float64)(dev.to_device(pb["x"]), dev.to_device(pb["cc"]), y,
                                                        dev.to_device(pb["detJ"]), dev.to_device(mesh.dofmap))
        finally:
            ops._MASS_PLAN_MIN_ENTRIES = old_min
        _check(y.copy_to_host(), y_ref, np.float64, f"planned mass, runs={runs} P={P}")
    finally:
        lib.set_tuning(lib.TUNE_PLAN_VARIANT, old)
        lib.set_tuning(lib.TUNE_PLAN_RUNS, 1)
        ops._PLANS.clear()


@pytest.mark.parametrize("path", golden_files("ops_"), ids=lambda p: p.split("/")[-1][:-4])
def test_device_precompute_vs_reference(gpu, path):
    """Device geometry kernels against what the reference's own precompute.py produced."""
    dev, _ = gpu
    pre = pkg("precompute")
    d = np.load(path)
    dt = d["x"].dtype
    tol = 1e-13 if dt == np.float64 else 5e-6
    nc = d["dofmap"].shape[0]
    mesh = (dev.to_device(d["x_dofs"]), dev.to_device(d["x_g"]))
    G, detJ = dev.device_array(d["ref_G"].shape, dt), dev.device_array(d["ref_detJ"].shape, dt)
    pre.compute_scaled_geometrical_factor_device(G, mesh, nc, dev.to_device(d["dphi_geom"]), dev.to_device(d["wts3"]), detJ=detJ)
    assert rel_l2(G.copy_to_host(), d["ref_G"]) < tol
    assert rel_l2(detJ.copy_to_host(), d["ref_detJ"]) < tol
    detJ2 = dev.device_array(d["ref_detJ"].shape, dt)
    pre.compute_scaled_jacobian_determinant_device(detJ2, mesh, nc, dev.to_device(d["dphi_geom"]), dev.to_device(d["wts3"]))
    assert rel_l2(detJ2.copy_to_host(), d["ref_detJ"]) < tol
    dF = dev.device_array(d["ref_detJ_f"].shape, dt)
    pre.compute_boundary_facets_scaled_jacobian_determinant_device(
        dF, mesh, dev.to_device(d["boundary_data"].astype(np.int32)), dev.to_device(d["dphi_facet"]), dev.to_device(d["wts2"]))
    assert rel_l2(dF.copy_to_host(), d["ref_detJ_f"]) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P", [2, 4, 6])
def test_cached_diagonal_mass_equals_mass_operator(gpu, oracle_c, P, dtype):
    """diagonal_mass_operator: y += (M(c) 1) (.) x equals the oracle's gather-scale-scatter mass apply (GLL collocation makes
    the operator diagonal), accumulate-into semantics included; refresh() follows a change of the constants."""
    import torch

    dev, ops = gpu
    pb = build_problem(P, 5, dtype=dtype, perturb=0.16)
    mesh = pb["mesh"]
    rng = np.random.default_rng(3)
    y0 = rng.standard_normal(mesh.ndofs).astype(dtype)
    y_ref = y0.astype(np.float64)
    oracle_c.mass_apply(pb["x"].astype(np.float64), pb["cc"].astype(np.float64), y_ref, pb["detJ"].astype(np.float64), mesh.dofmap)
    cc_d, dj_d, dm_d = dev.to_device(pb["cc"]), dev.to_device(pb["detJ"]), dev.to_device(mesh.dofmap)
    op = ops.diagonal_mass_operator(cc_d, dj_d, dm_d, mesh.ndofs, dtype)
    y = dev.to_device(y0.copy())
    op(dev.to_device(pb["x"]), y)
    torch.cuda.synchronize()
    _check(y.cpu().numpy(), y_ref, dtype, "cached-diagonal mass")
    cc_d.mul_(2.0)
    op.refresh()
    y2 = dev.to_device(np.zeros(mesh.ndofs, dtype=dtype))
    op(dev.to_device(pb["x"]), y2)
    torch.cuda.synchronize()
    _check(y2.cpu().numpy(), 2.0 * (y_ref - y0), dtype, "cached-diagonal mass after refresh")


def test_config2_full_size(gpu, oracle_c):
    """BASELINE config 2 at its stated size (P = 4, 25^3 perturbed cells, 1 030 301 dofs: the small-mesh regime,
    1.5 rounds of the chip): stiffness (planned and plan-free), cell mass (planned) and the boundary-facet mass of all
    six faces, each against the oracle, fp64 to 1e-12."""
    import torch

    dev, ops = gpu
    boxmesh, gll, pre = pkg("boxmesh"), pkg("gll"), pkg("precompute")
    P, N = 4, 25
    mesh = boxmesh.BoxMesh(P, N, perturb=0.16, seed=0)
    assert mesh.ndofs == 1030301 and mesh.ncells == 15625
    pts, wts, D = gll.tabulate_1d(P)
    n = P + 1
    gm = (dev.to_device(mesh.x_dofs), dev.to_device(mesh.x_g))
    dg = dev.to_device(pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts)))
    w3 = dev.to_device(gll.tensor_weights_3d(wts))
    G = dev.device_array((mesh.ncells, n**3, 6), np.float64)
    detJ = dev.device_array((mesh.ncells, n**3), np.float64)
    pre.compute_scaled_geometrical_factor_device(G, gm, mesh.ncells, dg, w3, detJ=detJ)
    bd = mesh.boundary_facets()
    assert bd.shape[0] == 6 * N * N
    dF = dev.device_array((bd.shape[0], n * n), np.float64)
    pre.compute_boundary_facets_scaled_jacobian_determinant_device(
        dF, gm, dev.to_device(bd.astype(np.int32)), dev.to_device(pre.tabulate_facet_gradients(pts)), dev.to_device(gll.tensor_weights_2d(wts)))
    fdm = mesh.facet_dofmap(bd)
    rng = np.random.default_rng(2)
    x = ref_field(mesh.dof_coordinates())
    cc = 1.0 + 0.25 * rng.standard_normal(mesh.ncells)
    fc = 1.0 + 0.25 * rng.standard_normal(bd.shape[0])
    x_d, cc_d, fc_d, dm, fdm_d = (dev.to_device(a) for a in (x, cc, fc, mesh.dofmap, fdm))
    threads = max(1, min(32, oracle_c.max_threads()))
    G_h, detJ_h, dF_h = G.copy_to_host(), detJ.copy_to_host(), dF.copy_to_host()
    # stiffness
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, D, x, cc, y_ref, G_h, mesh.dofmap, threads=threads)
    op = ops.stiffness_operator(P, D.flatten(), np.float64)
    for plan in (True, False):
        ops.use_plan(plan)
        try:
            y = torch.zeros(mesh.ndofs, dtype=torch.float64, device="cuda")
            op(x_d, cc_d, y, G, dm)
            _check(y.cpu().numpy(), y_ref, np.float64, f"stiffness, config 2, plan={plan}")
        finally:
            ops.use_plan(True)
    # cell mass (>= 32 k entries: the planned kernel) and facet mass (3 750 facets x 25 dofs: planned too)
    for name, consts, dj_d, dj_h, dmap_d, dmap_h, nper in (("cell mass", (cc, cc_d), detJ, detJ_h, dm, mesh.dofmap, n**3),
                                                              ("facet mass", (fc, fc_d), dF, dF_h, fdm_d, fdm, n * n)):
        y_ref = np.zeros(mesh.ndofs)
        oracle_c.mass_apply(x, consts[0], y_ref, dj_h, dmap_h)
        y = torch.zeros(mesh.ndofs, dtype=torch.float64, device="cuda")
        ops.mass_operator(nper, np.float64)(x_d, consts[1], y, dj_d, dmap_d)
        _check(y.cpu().numpy(), y_ref, np.float64, f"{name}, config 2")


def test_config1_default_kernels(gpu, oracle_c):
    """BASELINE config 1 at its stated size (numba-cpu/time_operators.py: P = 2, 18^3 cells, 50 653 dofs) on the SHIPPED
    default kernels -- planned stiffness, planned cell mass, facet mass of all six faces -- against the oracle
    (config 1 was exercised on the plan-free variants only)."""
    import torch

    dev, ops = gpu
    boxmesh, gll, pre = pkg("boxmesh"), pkg("gll"), pkg("precompute")
    P, N = 2, 18
    pb = build_problem(P, N, perturb=0.16)
    mesh = pb["mesh"]
    assert mesh.ndofs == 50653 and mesh.ncells == 5832
    n = P + 1
    assert ops._USE_PLAN
    x_d, cc_d, dm = dev.to_device(pb["x"]), dev.to_device(pb["cc"]), dev.to_device(mesh.dofmap)
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap, threads=4)
    y = torch.zeros(mesh.ndofs, dtype=torch.float64, device="cuda")
    ops.stiffness_operator(P, pb["D"].flatten(), np.float64)(x_d, cc_d, y, dev.to_device(pb["G"]), dm)
    _check(y.cpu().numpy(), y_ref, np.float64, "planned stiffness, config 1")
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.mass_apply(pb["x"], pb["cc"], y_ref, pb["detJ"], mesh.dofmap)
    y = torch.zeros(mesh.ndofs, dtype=torch.float64, device="cuda")
    ops.mass_operator(n**3, np.float64)(x_d, cc_d, y, dev.to_device(pb["detJ"]), dm)
    _check(y.cpu().numpy(), y_ref, np.float64, "planned cell mass, config 1")
    bd = mesh.boundary_facets()
    assert bd.shape[0] == 6 * N * N
    dF = np.zeros((bd.shape[0], n * n))
    pre.compute_boundary_facets_scaled_jacobian_determinant(dF, (mesh.x_dofs, mesh.x_g), bd, pre.tabulate_facet_gradients(pb["pts"]),
                                                            gll.tensor_weights_2d(pb["wts"]))
    fdm = mesh.facet_dofmap(bd)
    fc = 1.0 + 0.25 * np.random.default_rng(5).standard_normal(bd.shape[0])
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.mass_apply(pb["x"], fc, y_ref, dF, fdm)
    y = torch.zeros(mesh.ndofs, dtype=torch.float64, device="cuda")
    ops.mass_operator(n * n, np.float64)(x_d, dev.to_device(fc), y, dev.to_device(dF), dev.to_device(fdm))
    _check(y.cpu().numpy(), y_ref, np.float64, "facet mass, config 1")


def test_full_size_config3(gpu, oracle_c):
    """BASELINE config 3 at full size (P = 4, 54^3 perturbed cells, 10 218 313 dofs): the planned
    and the plan-free kernel against the oracle (all host cores), plus size-independent
    properties -- K 1 = 0, symmetry, linearity."""
    import torch

    dev, ops = gpu
    boxmesh, gll, pre = pkg("boxmesh"), pkg("gll"), pkg("precompute")
    P, N = 4, 54
    mesh = boxmesh.BoxMesh(P, N, perturb=0.16, seed=0)
    assert mesh.ndofs == 10218313 and mesh.ncells == 157464
    pts, wts, D = gll.tabulate_1d(P)
    gm = (dev.to_device(mesh.x_dofs), dev.to_device(mesh.x_g))
    G = dev.device_array((mesh.ncells, 125, 6), np.float64)
    pre.compute_scaled_geometrical_factor_device(
        G, gm, mesh.ncells, dev.to_device(pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))),
        dev.to_device(gll.tensor_weights_3d(wts)))
    rng = np.random.default_rng(0)
    x = rng.standard_normal(mesh.ndofs)
    v = rng.standard_normal(mesh.ndofs)
    cc = 1.0 + 0.25 * rng.standard_normal(mesh.ncells)
    dm, cc_d = dev.to_device(mesh.dofmap), dev.to_device(cc)
    op = ops.stiffness_operator(P, D.flatten(), np.float64)

    def K(vec, plan=True):
        ops.use_plan(plan)
        try:
            y = torch.zeros(mesh.ndofs, dtype=torch.float64, device="cuda")
            op(dev.to_device(vec), cc_d, y, G, dm)
            return y.cpu().numpy()
        finally:
            ops.use_plan(True)

    Kx = K(x)
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, D, x, cc, y_ref, G.copy_to_host(), mesh.dofmap, threads=max(1, min(32, oracle_c.max_threads())))
    _check(Kx, y_ref, np.float64, "planned kernel, full config 3")
    _check(K(x, plan=False), y_ref, np.float64, "plan-free kernel, full config 3")
    scale = np.max(np.abs(Kx))
    assert np.max(np.abs(K(np.full(mesh.ndofs, 2.5)))) < 1e-9 * scale  # K const = 0
    Kv = K(v)
    a, b = float(v @ Kx), float(x @ Kv)
    assert abs(a - b) < 1e-11 * max(abs(a), abs(b))  # symmetry
    _check(K(2.0 * x - 3.0 * v), 2.0 * Kx - 3.0 * Kv, np.float64, "linearity")


def test_full_size_config5_operators(gpu, oracle_c):
    """BASELINE config 5's operators at the size its step is quoted on (P = 6, 36^3 bowl-warped trilinear cells, 10 218 313 dofs),
    each DIRECTLY against the oracle on all host cores: the general-G stiffness apply, the stiffness
    apply with G formed in the kernel, and the Westervelt cell pass b += K(c3) u + K(c4) v in both forms.  The oracle reads the
    reference's G array (numba-cpu/precompute.py:115-163 conventions, formed on the device: pinned to 1e-13 by
    test_device_precompute_vs_reference)."""
    import torch

    dev, ops = gpu
    boxmesh, gll, pre = pkg("boxmesh"), pkg("gll"), pkg("precompute")
    P, N, L = 6, 36, 0.12

    def bowl(xg):  # bench.py's config-5 warp (benchlib/steps.py)
        out = xg.copy()
        yy, zz = xg[:, 1] / L - 0.5, xg[:, 2] / L - 0.5
        out[:, 0] = xg[:, 0] + 0.15 * (L / N) * 4 * (yy * yy + zz * zz) * (1.0 - xg[:, 0] / L)
        return out

    mesh = boxmesh.BoxMesh(P, N, length=(L, L, L), warp=bowl)
    assert mesh.ndofs == 10218313 and mesh.ncells == 46656
    pts, wts, D = gll.tabulate_1d(P)
    d = torch.device("cuda", 0)
    dm, xd, xg = (torch.from_numpy(a).to(d) for a in (mesh.dofmap, mesh.x_dofs, mesh.x_g))
    G = torch.empty((mesh.ncells, 343, 6), dtype=torch.float64, device=d)
    pre.compute_scaled_geometrical_factor_device(
        G, (xd, xg), mesh.ncells, torch.from_numpy(pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))).to(d),
        torch.from_numpy(gll.tensor_weights_3d(wts)).to(d))
    G_h = G.cpu().numpy()
    rng = np.random.default_rng(6)
    u, v = rng.standard_normal(mesh.ndofs), rng.standard_normal(mesh.ndofs)
    c3, c4 = 0.5 + rng.random(mesh.ncells), 0.5 + rng.random(mesh.ncells)
    threads = max(1, min(32, oracle_c.max_threads()))
    Ku = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, D, u, c3, Ku, G_h, mesh.dofmap, threads=threads)
    b_ref = Ku.copy()
    oracle_c.stiffness_apply(P, D, v, c4, b_ref, G_h, mesh.dofmap, threads=threads)
    u_d, v_d, c3_d, c4_d = (torch.from_numpy(a).to(d) for a in (u, v, c3, c4))
    y = torch.zeros(mesh.ndofs, dtype=torch.float64, device=d)
    ops.stiffness_operator(P, D.flatten(), np.float64)(u_d, c3_d, y, G, dm)
    _check(y.cpu().numpy(), Ku, np.float64, "general-G stiffness, P = 6 36^3 bowl")
    y.zero_()
    ops.stiffness_operator(P, D.flatten(), np.float64, geometry=(xd, xg, pts, wts))(u_d, c3_d, y, None, dm)
    _check(y.cpu().numpy(), Ku, np.float64, "in-kernel-geometry stiffness, P = 6 36^3 bowl")
    y.zero_()
    ops.westervelt_cell_operator(P, D.flatten(), np.float64).stiffness_only(u_d, v_d, c3_d, c4_d, y, G, dm)
    _check(y.cpu().numpy(), b_ref, np.float64, "Westervelt cell pass (G array), P = 6 36^3 bowl")
    y.zero_()
    ops.westervelt_cell_operator(P, D.flatten(), np.float64, geometry=(mesh.x_g, pts, wts)).stiffness_only(u_d, v_d, c3_d, c4_d, y, xd, dm)
    _check(y.cpu().numpy(), b_ref, np.float64, "Westervelt cell pass (in-kernel geometry), P = 6 36^3 bowl")


def test_plan_cache_follows_dofmap_changes(gpu, oracle_c):
    """The cached batch plan is keyed on the dofmap array's identity AND version: an in-place
    change of the dofmap must be seen by the next apply."""
    dev, ops = gpu
    P = 2
    pb = build_problem(P, (3, 3, 3), perturb=0.1)
    mesh = pb["mesh"]
    op = ops.stiffness_operator(P, pb["D"].flatten(), np.float64)
    x, cc, G = dev.to_device(pb["x"]), dev.to_device(pb["cc"]), dev.to_device(pb["G"])
    dm = dev.to_device(mesh.dofmap)
    y = dev.to_device(np.zeros(mesh.ndofs))
    op(x, cc, y, G, dm)
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap)
    _check(y.copy_to_host(), y_ref, np.float64, "first dofmap")
    perm = np.random.default_rng(0).permutation(mesh.ndofs).astype(np.int32)
    dm2 = perm[mesh.dofmap]  # renumbered dofs, same array object on the device
    dm.copy_(dev.to_device(dm2))
    y = dev.to_device(np.zeros(mesh.ndofs))
    op(x, cc, y, G, dm)
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], np.ascontiguousarray(dm2))
    _check(y.copy_to_host(), y_ref, np.float64, "dofmap changed in place")


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P", list(range(1, 11)))
def test_affine_fast_path(gpu, oracle_c, P, dtype):
    """Opt-in affine-cell path (reads G[c, 0, :] only) == general path on an affine box with
    anisotropic cells; the affinity check rejects a perturbed mesh."""
    dev, ops = gpu
    gll = pkg("gll")
    pb = build_problem(P, (4, 3, 5), dtype=dtype)  # h = (1/4, 1/3, 1/5): affine, anisotropic
    mesh = pb["mesh"]
    w3 = gll.tensor_weights_3d(pb["wts"])
    G = dev.to_device(pb["G"])
    assert ops.is_affine_geometry(G, w3, rtol=1e-12 if dtype == np.float64 else 1e-5)
    y_ref = np.zeros(mesh.ndofs, dtype=dtype)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap)
    y = dev.to_device(np.zeros(mesh.ndofs, dtype=dtype))
    op = ops.stiffness_operator(P, pb["D"].flatten(), dtype, affine_weights=w3)
    op(dev.to_device(pb["x"]), dev.to_device(pb["cc"]), y, G, dev.to_device(mesh.dofmap))
    _check(y.copy_to_host(), y_ref, dtype, f"affine fast path P={P}")
    pert = build_problem(P, (3, 2, 2), dtype=dtype, perturb=0.16)
    assert not ops.is_affine_geometry(dev.to_device(pert["G"]), w3, rtol=1e-12 if dtype == np.float64 else 1e-5)


@pytest.mark.parametrize("path", golden_files("ops_"), ids=lambda p: p.split("/")[-1][:-4])
def test_in_kernel_geometry_vs_reference(gpu, path):
    """Stiffness apply with G formed in the kernel from the 8 cell vertices (no G array) against the
    reference's own stiffness output, which consumed the G its precompute.py built from the same
    vertices (numba-cpu/precompute.py:115-163 -> operators.py:71-227)."""
    dev, ops = gpu
    d = np.load(path)
    P, dt = int(d["P"]), d["x"].dtype
    y = dev.to_device(d["y0"])
    op = ops.stiffness_operator(P, d["dphi_1d"].flatten(), dt, geometry=(d["x_dofs"], d["x_g"], d["pts"], d["wts"]))
    op(dev.to_device(d["x"]), dev.to_device(d["cell_constants"]), y, None, dev.to_device(d["dofmap"]))
    _check(y.copy_to_host(), d["ref_y_stiffness"], dt, "stiffness, in-kernel geometry")


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P", list(range(1, 11)))
def test_in_kernel_geometry_all_degrees(gpu, oracle_c, P, dtype):
    """P = 1..10, strongly perturbed trilinear cells, ragged last batch, random cell constants."""
    dev, ops = gpu
    pb = build_problem(P, (3, 2, 3) if P <= 6 else (2, 1, 2), dtype=dtype, perturb=0.25, seed=100 + P)
    mesh = pb["mesh"]
    y0 = np.random.default_rng(P).standard_normal(mesh.ndofs).astype(dtype)
    y_ref = y0.copy()
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap)
    y = dev.to_device(y0)
    op = ops.stiffness_operator(P, pb["D"].flatten(), dtype, geometry=(mesh.x_dofs, mesh.x_g, pb["pts"], pb["wts"]))
    op(dev.to_device(pb["x"]), dev.to_device(pb["cc"]), y, None, dev.to_device(mesh.dofmap))
    _check(y.copy_to_host(), y_ref, dtype, f"in-kernel geometry P={P}")


def test_in_kernel_geometry_full_size_properties(gpu, oracle_c):
    """BASELINE config 3 size (P = 4, 54^3 perturbed cells): size-independent properties of the
    in-kernel-geometry operator -- K 1 = 0, symmetry v.Ku = u.Kv, and agreement with the general-G
    kernel fed by the device precompute of the same vertices."""
    import torch

    dev, ops = gpu
    gll, boxmesh, pre = pkg("gll"), pkg("boxmesh"), pkg("precompute")
    P, N = 4, 54
    mesh = boxmesh.BoxMesh(P, N, perturb=0.16, seed=0)
    pts, wts, D = gll.tabulate_1d(P, np.float64)
    d = torch.device("cuda", 0)
    dm, xd, xg = (torch.from_numpy(a).to(d) for a in (mesh.dofmap, mesh.x_dofs, mesh.x_g))
    cc = torch.from_numpy(1.0 + 0.25 * np.random.default_rng(5).standard_normal(mesh.ncells)).to(d)
    opg = ops.stiffness_operator(P, D.flatten(), np.float64, geometry=(xd, xg, pts, wts))
    g = torch.Generator(device=d).manual_seed(3)
    u = torch.randn(mesh.ndofs, dtype=torch.float64, device=d, generator=g)
    v = torch.randn(mesh.ndofs, dtype=torch.float64, device=d, generator=g)
    one = torch.ones_like(u)
    Ku, Kv, K1 = torch.zeros_like(u), torch.zeros_like(u), torch.zeros_like(u)
    opg(u, cc, Ku, None, dm)
    opg(v, cc, Kv, None, dm)
    opg(one, cc, K1, None, dm)
    scale = float(Ku.abs().max())
    assert float(K1.abs().max()) < 1e-10 * scale
    a, b = float(torch.dot(v, Ku)), float(torch.dot(u, Kv))
    assert abs(a - b) < 1e-11 * max(abs(a), abs(b))
    G = torch.empty((mesh.ncells, (P + 1) ** 3, 6), dtype=torch.float64, device=d)
    pre.compute_scaled_geometrical_factor_device(
        G, (xd, xg), mesh.ncells, torch.from_numpy(pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))).to(d),
        torch.from_numpy(gll.tensor_weights_3d(wts)).to(d))
    Ku2 = torch.zeros_like(u)
    ops.stiffness_operator(P, D.flatten(), np.float64)(u, cc, Ku2, G, dm)
    err = float((Ku - Ku2).norm() / Ku2.norm())
    assert err < 1e-12, err
    # ... and DIRECTLY against the oracle at this size (round 6; until then only through the general-G GPU kernel above)
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, D, u.cpu().numpy(), cc.cpu().numpy(), y_ref, G.cpu().numpy(), mesh.dofmap, threads=max(1, min(32, oracle_c.max_threads())))
    _check(Ku.cpu().numpy(), y_ref, np.float64, "in-kernel geometry vs oracle, full config 3")


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 6, 8])
def test_random_cell_order_uses_locality_plan(gpu, oracle_c, P, dtype):
    """A dofmap whose rows are in random order: the plan cache builds the batches in min-dof order (an
    index indirection inside the plan -- G, detJ and the constants are NOT permuted) for every planned
    kernel: stiffness, in-kernel geometry, affine, cell mass, fused Westervelt pass."""
    dev, ops = gpu
    gll = pkg("gll")
    # more than two batches of cells, the last one ragged for P != 4 (its spare threads / missing cells load the last valid cell's lines)
    shape = {1: (8, 6, 5), 2: (5, 4, 5), 3: (5, 3, 5), 4: (4, 3, 5), 5: (3, 3, 4), 6: (3, 2, 3), 8: (7, 1, 2)}[P]
    pb = build_problem(P, shape, dtype=dtype, perturb=0.2, seed=21)
    mesh = pb["mesh"]
    perm = np.random.default_rng(3).permutation(mesh.ncells)
    dm, G, detJ, cc, xd = (np.ascontiguousarray(a[perm]) for a in (mesh.dofmap, pb["G"], pb["detJ"], pb["cc"], mesh.x_dofs))
    x_d, cc_d, dm_d, G_d, dJ_d = (dev.to_device(a) for a in (pb["x"], cc, dm, G, detJ))
    ops._PLANS.clear()
    # stiffness
    y_ref = np.zeros(mesh.ndofs, dtype=dtype)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], cc, y_ref, G, dm)
    y = dev.to_device(np.zeros(mesh.ndofs, dtype=dtype))
    ops.stiffness_operator(P, pb["D"].flatten(), dtype)(x_d, cc_d, y, G_d, dm_d)
    assert ops._PLANS.last_order is not None, "the random order must have triggered the locality plan"
    _check(y.copy_to_host(), y_ref, dtype, "stiffness, ordered plan")
    # in-kernel geometry
    y = dev.to_device(np.zeros(mesh.ndofs, dtype=dtype))
    ops.stiffness_operator(P, pb["D"].flatten(), dtype, geometry=(xd, mesh.x_g, pb["pts"], pb["wts"]))(x_d, cc_d, y, None, dm_d)
    _check(y.copy_to_host(), y_ref, dtype, "in-kernel geometry, ordered plan")
    # cell mass through the same plan
    y_ref = np.zeros(mesh.ndofs, dtype=dtype)
    oracle_c.mass_apply(pb["x"], cc, y_ref, detJ, dm)
    y = dev.to_device(np.zeros(mesh.ndofs, dtype=dtype))
    old_min, ops._MASS_PLAN_MIN_ENTRIES = ops._MASS_PLAN_MIN_ENTRIES, 1
    try:
        ops.mass_operator((P + 1) ** 3, dtype)(x_d, cc_d, y, dJ_d, dm_d)
    finally:
        ops._MASS_PLAN_MIN_ENTRIES = old_min
    _check(y.copy_to_host(), y_ref, dtype, "mass, ordered plan")
    # fused Westervelt pass: b += K(c3) u + K(c4) v + M(c5) v^2 ; m += M(c2) u
    rng = np.random.default_rng(8)
    v = rng.standard_normal(mesh.ndofs).astype(dtype)
    c2, c3, c4, c5 = ((0.5 + rng.random(mesh.ncells)).astype(dtype) for _ in range(4))
    b_ref, m_ref = np.zeros(mesh.ndofs, dtype=dtype), np.zeros(mesh.ndofs, dtype=dtype)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], c3, b_ref, G, dm)
    oracle_c.stiffness_apply(P, pb["D"], v, c4, b_ref, G, dm)
    oracle_c.mass_apply((v * v).astype(dtype), c5, b_ref, detJ, dm)
    oracle_c.mass_apply(pb["x"], c2, m_ref, detJ, dm)
    b, m = dev.to_device(np.zeros(mesh.ndofs, dtype=dtype)), dev.to_device(np.zeros(mesh.ndofs, dtype=dtype))
    ops.westervelt_cell_operator(P, pb["D"].flatten(), dtype)(x_d, dev.to_device(v), dev.to_device(c2), dev.to_device(c3),
                                                              dev.to_device(c4), dev.to_device(c5), b, m, G_d, dJ_d, dm_d)
    _check(b.copy_to_host(), b_ref, dtype, "Westervelt b, ordered plan")
    _check(m.copy_to_host(), m_ref, dtype, "Westervelt m, ordered plan")
    # ... and with the geometry formed in the kernel (vertex ids through the plan's order: row -> vertex id -> coordinate)
    b, m = dev.to_device(np.zeros(mesh.ndofs, dtype=dtype)), dev.to_device(np.zeros(mesh.ndofs, dtype=dtype))
    ops.westervelt_cell_operator(P, pb["D"].flatten(), dtype, geometry=(mesh.x_g, pb["pts"], pb["wts"]))(
        x_d, dev.to_device(v), dev.to_device(c2), dev.to_device(c3), dev.to_device(c4), dev.to_device(c5), b, m, dev.to_device(xd), dm_d)
    _check(b.copy_to_host(), b_ref, dtype, "Westervelt b, in-kernel geometry, ordered plan")
    _check(m.copy_to_host(), m_ref, dtype, "Westervelt m, in-kernel geometry, ordered plan")
    ops._PLANS.clear()


def test_affine_with_random_cell_order(gpu, oracle_c):
    dev, ops = gpu
    gll = pkg("gll")
    P = 3
    pb = build_problem(P, (4, 3, 5))
    mesh = pb["mesh"]
    perm = np.random.default_rng(5).permutation(mesh.ncells)
    dm, G, cc = (np.ascontiguousarray(a[perm]) for a in (mesh.dofmap, pb["G"], pb["cc"]))
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], cc, y_ref, G, dm)
    y = dev.to_device(np.zeros(mesh.ndofs))
    ops._PLANS.clear()
    op = ops.stiffness_operator(P, pb["D"].flatten(), np.float64, affine_weights=gll.tensor_weights_3d(pb["wts"]))
    op(dev.to_device(pb["x"]), dev.to_device(cc), y, dev.to_device(G), dev.to_device(dm))
    assert ops._PLANS.last_order is not None
    _check(y.copy_to_host(), y_ref, np.float64, "affine, ordered plan")


def test_planned_apply_rejects_foreign_workspace(gpu):
    """A workspace that was not built for this (degree, cell count) -- or not built at all -- is refused
    with FUS_ERR_PLAN_MISMATCH before any kernel indexes it."""
    import torch

    dev, ops = gpu
    lib = pkg("_lib").load()
    pb = build_problem(2, (3, 3, 3))
    mesh = pb["mesh"]
    d = torch.device("cuda", 0)
    x, cc, G, dm = (torch.from_numpy(a).to(d) for a in (pb["x"], pb["cc"], pb["G"], mesh.dofmap))
    y = torch.zeros_like(x)
    D = torch.from_numpy(pb["D"]).to(d)
    nbytes = lib.fus_stiffness_plan_bytes(2, mesh.ncells)
    ws = torch.zeros(int(nbytes) + 512, dtype=torch.uint8, device=d)
    base = ws.data_ptr() + (-ws.data_ptr() % 256)
    args = (x.data_ptr(), cc.data_ptr(), y.data_ptr(), G.data_ptr())
    assert lib.fus_stiffness_apply_planned_f64(*args, base, D.data_ptr(), 2, mesh.ncells, None) == -6  # never built
    assert lib.fus_stiffness_plan_build(dm.data_ptr(), 2, mesh.ncells, base, int(nbytes), None) == 0
    assert lib.fus_stiffness_apply_planned_f64(*args, base, D.data_ptr(), 2, mesh.ncells, None) == 0
    assert lib.fus_stiffness_apply_planned_f64(*args, base, D.data_ptr(), 2, mesh.ncells - 1, None) == -6  # other cell count
    assert lib.fus_stiffness_apply_planned_f64(*args, base, D.data_ptr(), 3, mesh.ncells, None) == -6  # other degree
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P", list(range(1, 11)))
def test_westervelt_cell_pass_in_kernel_geometry(gpu, oracle_c, P, dtype):
    """b += K(c3) u + K(c4) v + M(c5) v^2, m += M(c2) u with G and detJ formed in the kernel, against
    the four reference-style applies of the oracle on the precomputed G / detJ of the same vertices."""
    dev, ops = gpu
    pb = build_problem(P, (3, 2, 3) if P <= 6 else (2, 1, 2), dtype=dtype, perturb=0.25, seed=40 + P)
    mesh = pb["mesh"]
    rng = np.random.default_rng(P)
    v = rng.standard_normal(mesh.ndofs).astype(dtype)
    c2, c3, c4, c5 = ((0.5 + rng.random(mesh.ncells)).astype(dtype) for _ in range(4))
    b0, m0 = rng.standard_normal(mesh.ndofs).astype(dtype), rng.standard_normal(mesh.ndofs).astype(dtype)
    b_ref, m_ref = b0.copy(), m0.copy()
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], c3, b_ref, pb["G"], mesh.dofmap)
    oracle_c.stiffness_apply(P, pb["D"], v, c4, b_ref, pb["G"], mesh.dofmap)
    oracle_c.mass_apply((v * v).astype(dtype), c5, b_ref, pb["detJ"], mesh.dofmap)
    oracle_c.mass_apply(pb["x"], c2, m_ref, pb["detJ"], mesh.dofmap)
    b, m = dev.to_device(b0), dev.to_device(m0)
    op = ops.westervelt_cell_operator(P, pb["D"].flatten(), dtype, geometry=(mesh.x_g, pb["pts"], pb["wts"]))
    op(dev.to_device(pb["x"]), dev.to_device(v), dev.to_device(c2), dev.to_device(c3), dev.to_device(c4), dev.to_device(c5),
       b, m, dev.to_device(mesh.x_dofs), dev.to_device(mesh.dofmap))
    _check(b.copy_to_host(), b_ref, dtype, f"Westervelt b, in-kernel geometry P={P}")
    _check(m.copy_to_host(), m_ref, dtype, f"Westervelt m, in-kernel geometry P={P}")


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P", list(range(1, 11)))
def test_westervelt_stiffness_part_and_diagonal_mass(gpu, oracle_c, P, dtype):
    """The diagonal form of the Westervelt stage: (i) the cell pass with c2 = c5 = m = detJ = NULL is
    b += K(c3) u + K(c4) v (general G and in-kernel geometry); (ii) GLL collocation: M(c) x == diag(M(c) 1) x,
    the identity the solver's pointwise mass terms rest on, checked on the reference-pinned mass operator."""
    dev, ops = gpu
    pb = build_problem(P, (3, 2, 3), dtype=dtype, perturb=0.25, seed=60 + P)
    mesh = pb["mesh"]
    rng = np.random.default_rng(P)
    v = rng.standard_normal(mesh.ndofs).astype(dtype)
    c3, c4 = ((0.5 + rng.random(mesh.ncells)).astype(dtype) for _ in range(2))
    b_ref = np.zeros(mesh.ndofs, dtype=dtype)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], c3, b_ref, pb["G"], mesh.dofmap)
    oracle_c.stiffness_apply(P, pb["D"], v, c4, b_ref, pb["G"], mesh.dofmap)
    args = (dev.to_device(pb["x"]), dev.to_device(v), dev.to_device(c3), dev.to_device(c4))
    b = dev.to_device(np.zeros(mesh.ndofs, dtype=dtype))
    ops.westervelt_cell_operator(P, pb["D"].flatten(), dtype).stiffness_only(*args, b, dev.to_device(pb["G"]), dev.to_device(mesh.dofmap))
    _check(b.copy_to_host(), b_ref, dtype, f"K(c3)u + K(c4)v P={P}")
    b = dev.to_device(np.zeros(mesh.ndofs, dtype=dtype))
    opg = ops.westervelt_cell_operator(P, pb["D"].flatten(), dtype, geometry=(mesh.x_g, pb["pts"], pb["wts"]))
    opg.stiffness_only(*args, b, dev.to_device(mesh.x_dofs), dev.to_device(mesh.dofmap))
    _check(b.copy_to_host(), b_ref, dtype, f"K(c3)u + K(c4)v, in-kernel geometry P={P}")
    # (ii)
    ones = np.ones(mesh.ndofs, dtype=dtype)
    diag, full = np.zeros(mesh.ndofs, dtype=dtype), np.zeros(mesh.ndofs, dtype=dtype)
    oracle_c.mass_apply(ones, c3, diag, pb["detJ"], mesh.dofmap)
    oracle_c.mass_apply(v, c3, full, pb["detJ"], mesh.dofmap)
    _check(diag * v, full, dtype, "M(c) x == diag(M(c) 1) x")


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P", [1, 2, 3, 4, 5, 6, 8, 10])
def test_lists_and_run_tables_of_one_plan(gpu, oracle_c, P, dtype):
    """A plan holds the distinct-dof lists AND their run-length tables; which one a launch reads is a
    launch-time choice (auto: fp64 tables, fp32 lists).  All three settings on ONE cached plan, for the
    stiffness, in-kernel-geometry and mass kernels."""
    dev, ops = gpu
    lib = pkg("_lib")
    pb = build_problem(P, (5, 3, 4) if P < 6 else (3, 2, 3), dtype=dtype, perturb=0.16, seed=13)
    mesh = pb["mesh"]
    y_ref, m_ref = np.zeros(mesh.ndofs, dtype=dtype), np.zeros(mesh.ndofs, dtype=dtype)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap)
    oracle_c.mass_apply(pb["x"], pb["cc"], m_ref, pb["detJ"], mesh.dofmap)
    x, cc, G, dJ, dm = (dev.to_device(a) for a in (pb["x"], pb["cc"], pb["G"], pb["detJ"], mesh.dofmap))
    op = ops.stiffness_operator(P, pb["D"].flatten(), dtype)
    opg = ops.stiffness_operator(P, pb["D"].flatten(), dtype, geometry=(mesh.x_dofs, mesh.x_g, pb["pts"], pb["wts"]))
    ops._PLANS.clear()
    old_min, ops._MASS_PLAN_MIN_ENTRIES = ops._MASS_PLAN_MIN_ENTRIES, 1
    try:
        for mode in (1, 0, 2, 1):
            lib.set_tuning(lib.TUNE_PLAN_RUNS, mode)  # the plan is built at the first apply (mode 1: tables built)
            for fn, ref, what in ((lambda y: op(x, cc, y, G, dm), y_ref, "stiffness"), (lambda y: opg(x, cc, y, None, dm), y_ref, "geom"),
                                  (lambda y: ops.mass_operator((P + 1) ** 3, dtype)(x, cc, y, dJ, dm), m_ref, "mass")):
                y = dev.to_device(np.zeros(mesh.ndofs, dtype=dtype))
                fn(y)
                _check(y.copy_to_host(), ref, dtype, f"{what}, run-table mode {mode}, P={P}")
        # one row-ordered plan for the three kernels (+ the in-kernel-geometry operator's strip-ordered twin where it was kept)
        assert len([k for k in ops._PLANS._plans if k[-1] != "strips"]) == 1 and len(ops._PLANS._plans) <= 2
    finally:
        ops._MASS_PLAN_MIN_ENTRIES = old_min
        lib.set_tuning(lib.TUNE_PLAN_RUNS, 1)
        ops._PLANS.clear()


def test_plan_encoding_follows_the_numbering(gpu, oracle_c):
    """``fus_plan_encoding``: a lexicographic numbering compresses into run tables (every batch carries one, fp64 launches read them); a
    random renumbering of the dofs does not (no batch does), and the auto choice then reads the raw lists -- a run-coded launch would read
    every list one round trip late.  Same result as the oracle either way, and with the choice forced the other way."""
    import ctypes

    dev, ops = gpu
    lib = pkg("_lib")
    P = 4
    pb = build_problem(P, (6, 5, 7), perturb=0.16, seed=21)
    mesh = pb["mesh"]
    perm = np.random.default_rng(3).permutation(mesh.ndofs).astype(mesh.dofmap.dtype)
    clib = lib.load()
    for name, dofmap, xs in (("lexicographic", mesh.dofmap, pb["x"]), ("random", perm[mesh.dofmap], None)):
        if xs is None:  # x in the new numbering: x_new[perm[d]] = x[d]
            xs = np.empty_like(pb["x"])
            xs[perm] = pb["x"]
        y_ref = np.zeros(mesh.ndofs)
        oracle_c.stiffness_apply(P, pb["D"], xs, pb["cc"], y_ref, pb["G"], dofmap)
        x, cc, G, dm = (dev.to_device(a) for a in (xs, pb["cc"], pb["G"], np.ascontiguousarray(dofmap)))
        op = ops.stiffness_operator(P, pb["D"].flatten(), np.float64)
        ops._PLANS.clear()
        try:
            for mode in (1, 0, 2, 1):
                lib.set_tuning(lib.TUNE_PLAN_RUNS, mode)
                y = dev.to_device(np.zeros(mesh.ndofs))
                op(x, cc, y, G, dm)
                _check(y.copy_to_host(), y_ref, np.float64, f"{name} numbering, run-table mode {mode}")
                if mode == 1:
                    (ws, *_), = [v for k, v in ops._PLANS._plans.items() if k[-1] != "strips"]
                    nb, nr, r64, r32 = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
                    assert clib.fus_plan_encoding(ws.data_ptr(), ctypes.byref(nb), ctypes.byref(nr), ctypes.byref(r64), ctypes.byref(r32)) == 0
                    assert nb.value == -(-mesh.ncells // 10)
                    if name == "lexicographic":
                        assert nr.value == nb.value and r64.value == 1 and r32.value == 1
                    else:
                        assert nr.value == 0 and r64.value == 0 and r32.value == 0
        finally:
            lib.set_tuning(lib.TUNE_PLAN_RUNS, 1)
            ops._PLANS.clear()
    assert clib.fus_plan_encoding(0, None, None, None, None) != 0  # a workspace no plan was built in


def test_strip_ordered_plan_opt_in(gpu, oracle_c):
    """``use_strip_order(True)`` (opt-in, measured negative at config 3: docs/history.md section 11): the in-kernel-geometry operator's plan
    takes the two-row strip order when it lowers the distinct dofs per batch -- a second cached plan next to the row-ordered one, cell
    order inside the plan (no array moves), same result as the oracle."""
    import torch

    dev, ops = gpu
    P = 4
    pb = build_problem(P, (4, 6, 20), perturb=0.16, seed=5)
    mesh = pb["mesh"]
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.stiffness_apply(P, pb["D"], pb["x"], pb["cc"], y_ref, pb["G"], mesh.dofmap)
    x, cc, dm = (dev.to_device(a) for a in (pb["x"], pb["cc"], mesh.dofmap))
    opg = ops.stiffness_operator(P, pb["D"].flatten(), np.float64, geometry=(mesh.x_dofs, mesh.x_g, pb["pts"], pb["wts"]))
    ops._PLANS.clear()
    try:
        ops.use_strip_order(True)
        y = torch.zeros(mesh.ndofs, dtype=torch.float64, device="cuda")
        opg(x, cc, y, None, dm)
        _check(y.cpu().numpy(), y_ref, np.float64, "in-kernel geometry on a strip-ordered plan")
        keys = list(ops._PLANS._plans)
        assert len(keys) == 1 and keys[0][-1] == "strips" and ops._PLANS.last_order is not None  # the strip order was kept (-8 % distinct dofs)
        ops.use_strip_order(False)
        y2 = torch.zeros_like(y)
        opg(x, cc, y2, None, dm)
        _check(y2.cpu().numpy(), y_ref, np.float64, "in-kernel geometry on the row-ordered plan")
        assert len(ops._PLANS._plans) == 2
    finally:
        ops.use_strip_order(False)
        ops._PLANS.clear()


@pytest.mark.parametrize("path", golden_files("ops_P4_"), ids=lambda p: p.split("/")[-1][:-4])
def test_facet_terms_one_launch(gpu, oracle_c, path):
    """fus_facet_terms_*: the stage's boundary-facet mass applies (source facets with x = g filled into a
    vector -- and the dg term of the Westervelt solver -- plus the absorbing facets on v_n) in one launch,
    against the oracle's mass operator applied the reference's way on the golden facet arrays."""
    dev, ops = gpu
    d = np.load(path)
    dt = d["x"].dtype
    fdm, dJf, fc = d["bfacet_dofmap"], d["ref_detJ_f"], d["facet_constants"]
    half = fdm.shape[0] // 2
    A, B = slice(0, half), slice(half, None)  # "source" and "absorbing" facet sets
    g, dg = dt.type(0.37), dt.type(-1.9)
    c2 = (0.5 + np.random.default_rng(2).random(half)).astype(dt)
    v = d["x"]
    ref = d["y0"].copy()
    ones = np.ones_like(v)
    oracle_c.mass_apply((g * ones).astype(dt), np.ascontiguousarray(fc[A]), ref, np.ascontiguousarray(dJf[A]), np.ascontiguousarray(fdm[A]))
    oracle_c.mass_apply((dg * ones).astype(dt), c2, ref, np.ascontiguousarray(dJf[A]), np.ascontiguousarray(fdm[A]))
    oracle_c.mass_apply(v, np.ascontiguousarray(fc[B]), ref, np.ascontiguousarray(dJf[B]), np.ascontiguousarray(fdm[B]))
    y = dev.to_device(d["y0"])
    td = lambda a: dev.to_device(np.ascontiguousarray(a))  # noqa: E731
    ops.facet_terms(y, (td(fc[A]), float(g), td(c2), float(dg), td(dJf[A]), td(fdm[A])), (td(v), td(fc[B]), td(dJf[B]), td(fdm[B])))
    _check(y.copy_to_host(), ref, dt, "facet terms (two source constants + field)")
    # linear solver form: no second source constant; and an empty source set
    ref = d["y0"].copy()
    oracle_c.mass_apply((g * ones).astype(dt), np.ascontiguousarray(fc[A]), ref, np.ascontiguousarray(dJf[A]), np.ascontiguousarray(fdm[A]))
    oracle_c.mass_apply(v, np.ascontiguousarray(fc[B]), ref, np.ascontiguousarray(dJf[B]), np.ascontiguousarray(fdm[B]))
    y = dev.to_device(d["y0"])
    ops.facet_terms(y, (td(fc[A]), float(g), None, 0.0, td(dJf[A]), td(fdm[A])), (td(v), td(fc[B]), td(dJf[B]), td(fdm[B])))
    _check(y.copy_to_host(), ref, dt, "facet terms (linear solver form)")
    y2 = dev.to_device(d["y0"])
    ops.facet_terms(y2, (td(fc[:0]), 1.0, None, 0.0, td(dJf[:0]), td(fdm[:0])), (td(v), td(fc[B]), td(dJf[B]), td(fdm[B])))
    ref2 = d["y0"].copy()
    oracle_c.mass_apply(v, np.ascontiguousarray(fc[B]), ref2, np.ascontiguousarray(dJf[B]), np.ascontiguousarray(fdm[B]))
    _check(y2.copy_to_host(), ref2, dt, "facet terms (no source facets on this rank)")


@pytest.mark.timeout(900)
def test_maximum_size_beyond_32bit_offsets(gpu, oracle_c):
    """A single-GPU mesh whose G array has more than 2^31 ELEMENTS (P = 4, 143^3 = 2 924 207 perturbed
    cells, 188 M dofs, G = 17.5 GB): element offsets into G, the dofmap and the plan no longer fit 32 bits.
    Built on the device (the host mesh class would take minutes).  Checks: planned == plan-free; the whole
    launch == the sum of two launches over the cell halves (the second addressed through an offset base
    pointer, i.e. with small indices); the contribution of the LAST cells == the oracle; K 1 = 0; the
    in-kernel-geometry operator == the general one."""
    import torch

    dev, ops = gpu
    gll, pre = pkg("gll"), pkg("precompute")
    if torch.cuda.get_device_properties(0).total_memory < 100e9:
        pytest.skip("needs ~40 GB of device memory")
    d = torch.device("cuda", 0)
    P, N = 4, 143
    n, nd, vd = P + 1, P * N + 1, N + 1
    ncell, ndofs = N**3, (P * N + 1) ** 3
    assert ncell * n**3 * 6 > 2**31
    ar = torch.arange(N, device=d)
    cx, cy, cz = (t.reshape(-1) for t in torch.meshgrid(ar, ar, ar, indexing="ij"))
    li = torch.arange(n, device=d)
    I, J, K_ = (t.reshape(-1) for t in torch.meshgrid(li, li, li, indexing="ij"))
    dm = (((cx[:, None] * P + I[None, :]) * nd + (cy[:, None] * P + J[None, :])) * nd + (cz[:, None] * P + K_[None, :])).to(torch.int32)
    xd = torch.empty((ncell, 8), dtype=torch.int32, device=d)
    for v in range(8):  # basix P1 hex vertex order, as boxmesh.BoxMesh
        bx, by, bz = v & 1, (v >> 1) & 1, (v >> 2) & 1
        xd[:, v] = (((cx + bx) * vd + (cy + by)) * vd + (cz + bz)).to(torch.int32)
    del cx, cy, cz
    av = torch.arange(vd, device=d, dtype=torch.float64)
    h = 1.0 / N
    gen = torch.Generator(device=d).manual_seed(11)
    xg = torch.stack([t.reshape(-1) for t in torch.meshgrid(av, av, av, indexing="ij")], dim=1) * h
    xg += 0.16 * h * (2.0 * torch.rand(xg.shape, dtype=torch.float64, device=d, generator=gen) - 1.0)
    pts, wts, D = gll.tabulate_1d(P, np.float64)
    G = torch.empty((ncell, n**3, 6), dtype=torch.float64, device=d)
    pre.compute_scaled_geometrical_factor_device(
        G, (xd, xg), ncell, torch.from_numpy(pre.tabulate_hex_p1_gradients(gll.tensor_points_3d(pts))).to(d),
        torch.from_numpy(gll.tensor_weights_3d(wts)).to(d))
    cc = 1.0 + 0.25 * torch.randn(ncell, dtype=torch.float64, device=d, generator=gen)
    x = torch.randn(ndofs, dtype=torch.float64, device=d, generator=gen)
    op = ops.stiffness_operator(P, D.flatten(), np.float64)

    def K(vec, a=0, b=ncell, plan=True):
        ops.use_plan(plan)
        try:
            y = torch.zeros(ndofs, dtype=torch.float64, device=d)
            op(vec, cc[a:b], y, G[a:b], dm[a:b])
            return y
        finally:
            ops.use_plan(True)

    y_full = K(x)
    nrm = float(y_full.norm())
    scale = float(y_full.abs().max())
    assert np.isfinite(nrm) and nrm > 0
    assert float((K(x, plan=False) - y_full).norm()) < 1e-12 * nrm
    half = (ncell // 2 // 10) * 10 + 3  # not a multiple of the batch size
    y_parts = K(x, 0, half)
    y_parts += K(x, half, ncell)
    assert float((y_parts - y_full).norm()) < 1e-12 * nrm
    # the last cells (offsets beyond 2^31 elements in the full launch) against the oracle
    ntail = 3000
    y_tail = (y_full - K(x, 0, ncell - ntail)).cpu().numpy()
    ref = np.zeros(ndofs)
    oracle_c.stiffness_apply(P, D, x.cpu().numpy(), cc[ncell - ntail:].cpu().numpy(), ref, G[ncell - ntail:].cpu().numpy(),
                             dm[ncell - ntail:].cpu().numpy(), threads=1)
    assert np.max(np.abs(y_tail - ref)) < 1e-11 * scale
    assert float(K(torch.full_like(x, 2.5)).abs().max()) < 1e-9 * scale  # K const = 0
    del y_parts
    opg = ops.stiffness_operator(P, D.flatten(), np.float64, geometry=(xd, xg, pts, wts))
    y_g = torch.zeros_like(x)
    opg(x, cc, y_g, None, dm)
    assert float((y_g - y_full).norm()) < 1e-11 * nrm
    ops._PLANS.clear()


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P,cells", [(4, 12), (2, 20), (6, (6, 5, 4))], ids=["P4", "P2", "P6"])
def test_mass_exclusive_dof_marks(gpu, oracle_c, P, cells, dtype):
    """``mass_operator(N, T, exclusive=True, atomic=True)`` (the float-atomic batch plan with exclusive-dof marks): dofs that exactly one batch of the plan touches are finished with a plain load +
    store instead of an atomic.  Same result as the oracle; accumulates into y like the unmarked apply (two launches back to
    back on one stream); cell and facet entities; with every dof declared as used elsewhere nothing is marked."""
    import torch

    dev, ops = gpu
    pb = build_problem(P, cells, dtype=dtype, perturb=0.16)
    mesh = pb["mesh"]
    n = P + 1
    x64, cc64, dj64 = pb["x"].astype(np.float64), pb["cc"].astype(np.float64), pb["detJ"].astype(np.float64)
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.mass_apply(x64, cc64, y_ref, dj64, mesh.dofmap)
    x_d, cc_d, dj_d, dm = (dev.to_device(a) for a in (pb["x"], pb["cc"], pb["detJ"], mesh.dofmap))
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    op = ops.mass_operator(n**3, dtype, exclusive=True, atomic=True)
    y = torch.zeros(mesh.ndofs, dtype=tdt, device="cuda")
    op(x_d, cc_d, y, dj_d, dm)
    _check(y.cpu().numpy(), y_ref, dtype, "exclusive-marks cell mass")
    op(x_d, cc_d, y, dj_d, dm)  # accumulates: the plain load of the second launch sees the first launch's store
    _check(y.cpu().numpy(), 2 * y_ref, dtype, "exclusive-marks cell mass, second launch")
    # the marks themselves: some dofs are exclusive (cell interiors at least), the batches' shared faces are not
    ws, epb = ops._PLANS.get(dm, exclusive_ndofs=mesh.ndofs)
    lib = pkg("_lib").load()
    nbatch = (mesh.ncells + epb - 1) // epb
    nbytes = int(lib.fus_plan_bytes(n**3, epb, mesh.ncells))
    words = (epb * n**3 + 31) // 32
    excl = ws[nbytes - ((nbatch * words * 4 + 255) // 256 * 256):][: nbatch * words * 4].view(torch.int32).cpu().numpy().view(np.uint32)
    marked = int(sum(bin(int(w)).count("1") for w in excl))
    nu = (ws[256:256 + 4 * nbatch].view(torch.int32).cpu().numpy() & 0xFFFF).sum()
    assert marked >= mesh.ncells * (P - 1) ** 3 and marked < nu  # every cell-interior dof; not the shared faces
    # everything declared as used elsewhere: no marks, same result
    ext = torch.ones(mesh.ndofs, dtype=torch.int32, device="cuda")
    ws2, _ = ops._PLANS.get(dm, exclusive_ndofs=mesh.ndofs, external_use=ext)
    excl2 = ws2[nbytes - ((nbatch * words * 4 + 255) // 256 * 256):][: nbatch * words * 4]
    assert int(excl2.to(torch.int64).sum().item()) == 0
    # boundary facets (N = n^2)
    gll, pre = pkg("gll"), pkg("precompute")
    bd = mesh.boundary_facets()
    dF = np.zeros((bd.shape[0], n * n))
    pre.compute_boundary_facets_scaled_jacobian_determinant(dF, (mesh.x_dofs.astype(np.int32), mesh.x_g.astype(np.float64)), bd,
                                                            pre.tabulate_facet_gradients(gll.gll_points_weights(P)[0]),
                                                            gll.tensor_weights_2d(gll.gll_points_weights(P)[1]))
    fdm = mesh.facet_dofmap(bd)
    fc = 1.0 + 0.25 * np.random.default_rng(5).standard_normal(bd.shape[0])
    if bd.shape[0] * n * n >= ops._MASS_PLAN_MIN_ENTRIES:
        y_ref = np.zeros(mesh.ndofs)
        oracle_c.mass_apply(x64, fc, y_ref, dF, fdm)
        y = torch.zeros(mesh.ndofs, dtype=tdt, device="cuda")
        ops.mass_operator(n * n, dtype, exclusive=True, atomic=True)(x_d, dev.to_device(fc.astype(dtype)), y, dev.to_device(dF.astype(dtype)), dev.to_device(fdm))
        _check(y.cpu().numpy(), y_ref, dtype, "exclusive-marks facet mass")


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("P,cells,order", [(4, 12, "lex"), (3, (9, 9, 7), "random"), (6, (6, 5, 4), "lex"), (5, 7, "random"), (8, 4, "lex")],
                         ids=["P4", "P3-random-cell-order", "P6", "P5-random-cell-order", "P8"])
def test_mass_gather_kernel(gpu, oracle_c, P, cells, order, dtype):
    """The default cell mass apply: the atomic-free transposed-dofmap kernel (csrc/mass_gather.hpp).  Same result as the
    oracle (numba-cpu/operators.py:19-68 restated) and as the float-atomic kernel; accumulates into y (two launches back to
    back); bitwise reproducible from run to run; any cell order; boundary facets (touched dofs are a sparse subset: the plan
    carries its row list); the operator's ``.atomic`` twin launches the batch-plan kernel."""
    import torch

    dev, ops = gpu
    pb = build_problem(P, cells, dtype=dtype, perturb=0.16)
    mesh = pb["mesh"]
    n = P + 1
    perm = np.arange(mesh.ncells) if order == "lex" else np.random.default_rng(11).permutation(mesh.ncells)
    dofmap = np.ascontiguousarray(mesh.dofmap[perm])
    x64, cc64, dj64 = pb["x"].astype(np.float64), pb["cc"].astype(np.float64)[perm], pb["detJ"].astype(np.float64)[perm]
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.mass_apply(x64, cc64, y_ref, dj64, dofmap)
    x_d, cc_d, dj_d, dm = (dev.to_device(a) for a in (pb["x"], pb["cc"][perm], pb["detJ"][perm], dofmap))
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    assert mesh.ncells * n**3 >= ops._MASS_PLAN_MIN_ENTRIES
    assert ops.mass_kernel_name(dm, mesh.ndofs) == "fus::mass_gather_kernel"
    assert ops.mass_kernel_name(dm, mesh.ndofs, atomic=True) == "fus::mass_plan_kernel"
    op = ops.mass_operator(n**3, dtype)
    y = torch.zeros(mesh.ndofs, dtype=tdt, device="cuda")
    op(x_d, cc_d, y, dj_d, dm)
    _check(y.cpu().numpy(), y_ref, dtype, "gather cell mass")
    y1 = y.clone()
    op(x_d, cc_d, y, dj_d, dm)
    _check(y.cpu().numpy(), 2 * y_ref, dtype, "gather cell mass, second launch")
    y2 = torch.zeros_like(y)
    op(x_d, cc_d, y2, dj_d, dm)
    assert torch.equal(y1, y2), "the gather kernel must be bitwise reproducible"
    ya = torch.zeros_like(y)
    op.atomic(x_d, cc_d, ya, dj_d, dm)
    _check(ya.cpu().numpy(), y_ref, dtype, "atomic twin")
    info = ops._GATHER_PLANS.get(dm, mesh.ndofs)[2]
    assert info[0] == mesh.ndofs and info[1] == 1 and info[2] == 8  # every dof touched, rows 0..ndofs-1, vertex dofs in 8 cells
    # static_detJ=True: detJ streamed from a row-ordered copy (fus_mass_gather_static_*): the same sums in the same order
    ops._STATIC_DETJ.clear()
    ops_static = ops.mass_operator(n**3, dtype, static_detJ=True)
    ys = torch.zeros_like(y)
    ops_static(x_d, cc_d, ys, dj_d, dm)
    assert torch.equal(ys, y1), "static-detJ apply must equal the default apply bitwise"
    assert len([v for v in ops._STATIC_DETJ._entries.values() if v is not None]) == 1  # ... and it really took the static path
    cc2 = cc_d * 3.0  # the constants may change per apply
    ys2, yd2 = torch.zeros_like(y), torch.zeros_like(y)
    ops_static(x_d, cc2, ys2, dj_d, dm)
    op(x_d, cc2, yd2, dj_d, dm)
    assert torch.equal(ys2, yd2)
    dj_d.mul_(2.0)  # a torch in-place change of detJ is noticed (version counter): the copy is rebuilt
    ys3 = torch.zeros_like(y)
    ops_static(x_d, cc_d, ys3, dj_d, dm)
    _check(ys3.cpu().numpy(), 2 * y_ref, dtype, "static-detJ apply after detJ changed")
    dj_d.mul_(0.5)
    # every build of the kernel (rows per thread)
    lib = pkg("_lib")
    for variant in (1, 2, 4):
        lib.set_tuning(lib.TUNE_MASS_VARIANT, variant)
        try:
            yv = torch.zeros_like(y)
            op(x_d, cc_d, yv, dj_d, dm)
            assert torch.equal(yv, y1), f"variant {variant}"
            yv.zero_()
            ops_static(x_d, cc_d, yv, dj_d, dm)
            assert torch.equal(yv, y1), f"static, variant {variant}"
        finally:
            lib.set_tuning(lib.TUNE_MASS_VARIANT, 0)
    # boundary facets (N = n^2): only the boundary dofs are touched
    gll, pre = pkg("gll"), pkg("precompute")
    bd = mesh.boundary_facets()
    dF = np.zeros((bd.shape[0], n * n))
    pre.compute_boundary_facets_scaled_jacobian_determinant(dF, (mesh.x_dofs.astype(np.int32), mesh.x_g.astype(np.float64)), bd,
                                                            pre.tabulate_facet_gradients(gll.gll_points_weights(P)[0]),
                                                            gll.tensor_weights_2d(gll.gll_points_weights(P)[1]))
    fdm = mesh.facet_dofmap(bd)
    fc = 1.0 + 0.25 * np.random.default_rng(5).standard_normal(bd.shape[0])
    old = ops._MASS_PLAN_MIN_ENTRIES
    ops._MASS_PLAN_MIN_ENTRIES = 0  # small facet sets take the plan-free kernel by default: force the gather here
    try:
        fdm_d = dev.to_device(fdm)
        assert ops.mass_kernel_name(fdm_d, mesh.ndofs) == "fus::mass_gather_kernel"
        y_ref = np.full(mesh.ndofs, 0.5)
        oracle_c.mass_apply(x64, fc, y_ref, dF, fdm)
        y = torch.full((mesh.ndofs,), 0.5, dtype=tdt, device="cuda")  # untouched dofs keep their values
        ops.mass_operator(n * n, dtype)(x_d, dev.to_device(fc.astype(dtype)), y, dev.to_device(dF.astype(dtype)), fdm_d)
        _check(y.cpu().numpy(), y_ref, dtype, "gather facet mass")
        yfs = torch.full((mesh.ndofs,), 0.5, dtype=tdt, device="cuda")
        ops.mass_operator(n * n, dtype, static_detJ=True)(x_d, dev.to_device(fc.astype(dtype)), yfs, dev.to_device(dF.astype(dtype)), fdm_d)
        assert torch.equal(yfs, y), "static-detJ facet mass (sparse row list)"
        finfo = ops._GATHER_PLANS.get(fdm_d, mesh.ndofs)[2]
        assert finfo[0] == np.unique(fdm).size and finfo[1] == 0
    finally:
        ops._MASS_PLAN_MIN_ENTRIES = old
    ops._GATHER_PLANS.clear()


@pytest.mark.parametrize("seed", range(6))
def test_mass_gather_random_dofmaps(gpu, seed):
    """The transposed-dofmap mass apply on dofmaps no mesh produces: any entity size (1 ... 64), dofs repeated inside an
    entity, hot dofs with up to 250 entries next to untouched ones (row list, run lengths 1 ... 250, multi-batch rows),
    fp64 and fp32, every build of the kernel -- against numpy's add.at in float64."""
    import ctypes as C

    import torch

    lib = pkg("_lib")
    L = lib.load()
    rng = np.random.default_rng(100 + seed)
    N = int(rng.choice([1, 3, 8, 27, 64]))
    nent = int(rng.integers(1, 4000))
    nd = int(rng.integers(5, 20000))
    dm = rng.integers(0, nd, size=(nent, N))
    # a few hot dofs (at most 250 entries each, the plan's limit is 255) and a dead zone nobody touches
    hot = rng.choice(nd, size=min(3, nd), replace=False)
    flat = dm.reshape(-1)
    for h in hot:
        idx = rng.choice(flat.size, size=min(80, flat.size), replace=False)
        flat[idx] = h
    lo = nd // 3
    flat[(flat >= lo) & (flat < lo + nd // 10)] = 0 if seed % 2 else lo  # empties a range of dofs
    counts = np.bincount(flat, minlength=nd)
    if counts.max() > 255:  # keep inside the plan's limit: spread the excess
        for d in np.nonzero(counts > 250)[0]:
            where = np.nonzero(flat == d)[0][250:]
            flat[where] = rng.integers(lo + nd // 10, nd, size=where.size) if lo + nd // 10 < nd else d
        counts = np.bincount(flat, minlength=nd)
    assume_ok = counts.max() <= 255
    dm = flat.reshape(nent, N).astype(np.int32)
    x = rng.standard_normal(nd)
    c = rng.standard_normal(nent)
    dj = rng.uniform(0.5, 1.5, size=(nent, N))
    y0 = rng.standard_normal(nd)
    ref = y0.copy()
    np.add.at(ref, dm.reshape(-1), (x[dm] * dj * c[:, None]).reshape(-1))
    dm_d = torch.from_numpy(dm).cuda()
    nbytes = int(L.fus_mass_gather_plan_bytes(N, nent, nd))
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    rc = L.fus_mass_gather_plan_build(dm_d.data_ptr(), N, nent, nd, ws.data_ptr(), nbytes, None)
    if not assume_ok:
        assert rc == lib.ERR_UNSUPPORTED_ENTITY
        return
    assert rc == 0
    info = (C.c_int64 * 4)()
    assert L.fus_mass_gather_plan_info(ws.data_ptr(), info) == 0
    assert info[0] == int((counts > 0).sum()) and info[2] == int(counts.max())
    assert info[1] == int(bool((counts[: info[0]] > 0).all()))  # dense iff the touched dofs are 0 .. touched - 1
    try:
        for dt, tol in ((np.float64, 1e-13), (np.float32, 2e-5)):
            fn = getattr(L, f"fus_mass_apply_gather_{'f64' if dt == np.float64 else 'f32'}")
            xd, cd, djd = (torch.from_numpy(a.astype(dt)).cuda() for a in (x, c, dj))
            for variant in (0, 1, 2, 4):
                lib.set_tuning(lib.TUNE_MASS_VARIANT, variant)
                y = torch.from_numpy(y0.astype(dt)).cuda()
                assert fn(xd.data_ptr(), cd.data_ptr(), y.data_ptr(), djd.data_ptr(), ws.data_ptr(), N, nent, None) == 0
                torch.cuda.synchronize()
                err = np.abs(y.cpu().numpy().astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1.0)
                assert err < tol * max(1, int(counts.max())), f"N={N} nent={nent} nd={nd} {dt.__name__} variant {variant}: {err}"
                # the static companion (detJ in row order, 16-bit entity offsets): bitwise the same sums
                sbytes = int(L.fus_mass_gather_static_bytes(N, nent, 8 if dt == np.float64 else 4))
                sws = torch.empty(sbytes, dtype=torch.uint8, device="cuda")
                sfx = "f64" if dt == np.float64 else "f32"
                assert getattr(L, f"fus_mass_gather_static_build_{sfx}")(ws.data_ptr(), djd.data_ptr(), sws.data_ptr(), sbytes, None) == 0
                ys = torch.from_numpy(y0.astype(dt)).cuda()
                assert getattr(L, f"fus_mass_apply_gather_static_{sfx}")(xd.data_ptr(), cd.data_ptr(), ys.data_ptr(), ws.data_ptr(), sws.data_ptr(), N, nent, None) == 0
                torch.cuda.synchronize()
                assert torch.equal(ys, y), f"static companion, {dt.__name__} variant {variant}"
                other = "f32" if dt == np.float64 else "f64"  # a companion built for another element size is refused
                assert getattr(L, f"fus_mass_apply_gather_static_{other}")(xd.data_ptr(), cd.data_ptr(), ys.data_ptr(), ws.data_ptr(), sws.data_ptr(), N, nent, None) == -6
                L.fus_plan_release(sws.data_ptr())
                assert getattr(L, f"fus_mass_apply_gather_static_{sfx}")(xd.data_ptr(), cd.data_ptr(), ys.data_ptr(), ws.data_ptr(), sws.data_ptr(), N, nent, None) == -6
        # row subsets (the partitioned apply's split): two plans over disjoint dof sets add up to the full apply, each touches
        # only its own rows
        marks = torch.from_numpy((rng.random(nd) < 0.4).astype(np.uint8)).cuda()
        ws_ab = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(2)]
        for w, wsub in enumerate(ws_ab):
            assert L.fus_mass_gather_plan_build_rows(dm_d.data_ptr(), N, nent, nd, marks.data_ptr(), w, wsub.data_ptr(), nbytes, None) == 0
        xd, cd, djd = (torch.from_numpy(a).cuda() for a in (x, c, dj))
        mk = marks.cpu().numpy().astype(bool)
        for w, wsub in enumerate(ws_ab):
            assert L.fus_mass_gather_plan_info(wsub.data_ptr(), info) == 0
            assert info[0] == int(((counts > 0) & (mk == bool(w))).sum())
            y = torch.from_numpy(y0).cuda()
            assert L.fus_mass_apply_gather_f64(xd.data_ptr(), cd.data_ptr(), y.data_ptr(), djd.data_ptr(), wsub.data_ptr(), N, nent, None) == 0
            torch.cuda.synchronize()
            got = y.cpu().numpy()
            mine = mk == bool(w)
            assert np.array_equal(got[~mine], y0[~mine]), "a row-subset launch must not touch the other rows"
            assert np.abs(got[mine] - ref[mine]).max(initial=0.0) < 1e-13 * max(1, int(counts.max())) * max(np.abs(ref).max(), 1.0)
        assert L.fus_mass_gather_plan_build_rows(dm_d.data_ptr(), N, nent, nd, None, 0, ws_ab[0].data_ptr(), nbytes, None) == -1
    finally:
        lib.set_tuning(lib.TUNE_MASS_VARIANT, 0)
        L.fus_plan_release(ws.data_ptr())
        for wsub in locals().get("ws_ab", []):
            L.fus_plan_release(wsub.data_ptr())


def test_mass_gather_policy_and_errors(gpu, oracle_c):
    """P = 2 (27 / 8 = 3.4 entries per dof: the gather loses there) keeps the atomic batch plan by itself; the C ABI refuses
    what the plan cannot hold (a dofmap value outside the vector, a dof in more than 255 entities, 2^31 entries) and an apply
    with a workspace it did not build; FUS_MASS_GATHER's switch."""
    import ctypes as C

    import torch

    dev, ops = gpu
    lib = pkg("_lib")
    L = lib.load()
    pb = build_problem(2, 12, perturb=0.1)
    mesh = pb["mesh"]
    dm = dev.to_device(mesh.dofmap)
    assert mesh.ncells * 27 >= ops._MASS_PLAN_MIN_ENTRIES
    assert ops.mass_kernel_name(dm, mesh.ndofs) == "fus::mass_plan_kernel"
    x_d, cc_d, dj_d = (dev.to_device(a) for a in (pb["x"], pb["cc"], pb["detJ"]))
    y = torch.zeros(mesh.ndofs, dtype=torch.float64, device="cuda")
    ops.mass_operator(27, np.float64)(x_d, cc_d, y, dj_d, dm)
    y_ref = np.zeros(mesh.ndofs)
    oracle_c.mass_apply(pb["x"], pb["cc"], y_ref, pb["detJ"], mesh.dofmap)
    _check(y.cpu().numpy(), y_ref, np.float64, "P = 2 cell mass (atomic plan by policy)")
    # round 6: the STATIC-detJ form pays at P = 2 too (0.60 against 0.46 of the roofline at 10 M dofs, profiles/r06g_ab_mass_low_degree.log):
    # static_detJ=True takes the transposed dofmap + the row-ordered detJ there, the default operator keeps the atomic plan
    ys = torch.zeros_like(y)
    ops.mass_operator(27, np.float64, static_detJ=True)(x_d, cc_d, ys, dj_d, dm)
    _check(ys.cpu().numpy(), y_ref, np.float64, "P = 2 cell mass, static-detJ gather")
    assert len([v for v in ops._STATIC_DETJ._entries.values() if v is not None]) >= 1, "static_detJ=True must take the static gather path at P = 2"
    assert ops.mass_kernel_name(dm, mesh.ndofs) == "fus::mass_plan_kernel"  # (the plan kept for the static form does not change the default)
    ys2 = torch.zeros_like(y)
    ops.mass_operator(27, np.float64, static_detJ=True)(x_d, cc_d, ys2, dj_d, dm)
    assert torch.equal(ys, ys2), "the atomic-free kernel is bitwise reproducible"
    # ... but the kernel itself is right there too (policy threshold lifted)
    old = ops._GATHER_MAX_MEAN_ENTRIES
    ops._GATHER_MAX_MEAN_ENTRIES = 100.0
    ops._GATHER_PLANS.clear()
    try:
        assert ops.mass_kernel_name(dm, mesh.ndofs) == "fus::mass_gather_kernel"
        y.zero_()
        ops.mass_operator(27, np.float64)(x_d, cc_d, y, dj_d, dm)
        _check(y.cpu().numpy(), y_ref, np.float64, "P = 2 cell mass (gather)")
    finally:
        ops._GATHER_MAX_MEAN_ENTRIES = old
        ops._GATHER_PLANS.clear()
    # switch
    pb4 = build_problem(4, 7, perturb=0.1)
    dm4 = dev.to_device(pb4["mesh"].dofmap)
    ops.use_mass_gather(False)
    try:
        assert ops.mass_kernel_name(dm4, pb4["mesh"].ndofs) == "fus::mass_plan_kernel"
    finally:
        ops.use_mass_gather(True)
    assert ops.mass_kernel_name(dm4, pb4["mesh"].ndofs) == "fus::mass_gather_kernel"
    # C ABI
    assert L.fus_mass_gather_plan_bytes(0, 10, 10) == -1 and L.fus_mass_gather_plan_bytes(8, -1, 10) == -1
    assert L.fus_mass_gather_plan_bytes(125, 2**31 // 125 + 1, 1000) < 0  # 2^31 entries
    nent, N, nd = 300, 8, 50
    nbytes = int(L.fus_mass_gather_plan_bytes(N, nent, nd))
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    bad = torch.zeros((nent, N), dtype=torch.int32, device="cuda")  # dof 0 in 2400 entries
    assert L.fus_mass_gather_plan_build(bad.data_ptr(), N, nent, nd, ws.data_ptr(), nbytes, None) == lib.ERR_UNSUPPORTED_ENTITY
    oob = torch.arange(nent * N, dtype=torch.int32, device="cuda").reshape(nent, N) % nd
    oob[5, 3] = nd  # one past the vector
    assert L.fus_mass_gather_plan_build(oob.data_ptr(), N, nent, nd, ws.data_ptr(), nbytes, None) == lib.ERR_UNSUPPORTED_ENTITY
    oob[5, 3] = -1
    assert L.fus_mass_gather_plan_build(oob.data_ptr(), N, nent, nd, ws.data_ptr(), nbytes, None) == lib.ERR_UNSUPPORTED_ENTITY
    v = torch.zeros(nd, dtype=torch.float64, device="cuda")
    c = torch.ones(nent, dtype=torch.float64, device="cuda")
    dj = torch.ones((nent, N), dtype=torch.float64, device="cuda")
    rc = L.fus_mass_apply_gather_f64(v.data_ptr(), c.data_ptr(), v.data_ptr(), dj.data_ptr(), ws.data_ptr(), N, nent, None)
    assert rc == -6  # FUS_ERR_PLAN_MISMATCH: nothing was registered at this address
    info = (C.c_int64 * 4)()
    assert L.fus_mass_gather_plan_info(ws.data_ptr(), info) == -6
    oob[5, 3] = 7
    assert L.fus_mass_gather_plan_build(oob.data_ptr(), N, nent, nd, ws.data_ptr(), nbytes, None) == 0
    assert L.fus_mass_apply_gather_f64(v.data_ptr(), c.data_ptr(), v.data_ptr(), dj.data_ptr(), ws.data_ptr(), N + 1, nent, None) == -6
    xx = torch.arange(1, nd + 1, dtype=torch.float64, device="cuda")
    assert L.fus_mass_apply_gather_f64(xx.data_ptr(), c.data_ptr(), v.data_ptr(), dj.data_ptr(), ws.data_ptr(), N, nent, None) == 0
    torch.cuda.synchronize()
    cnt = np.bincount(oob.cpu().numpy().reshape(-1), minlength=nd)
    assert np.allclose(v.cpu().numpy(), cnt * np.arange(1, nd + 1))
    assert L.fus_plan_release(ws.data_ptr()) == 0
    assert L.fus_mass_gather_plan_info(ws.data_ptr(), info) == -6
    # empty entity set
    e0 = int(L.fus_mass_gather_plan_bytes(N, 0, nd))
    ws0 = torch.empty(max(e0, 256), dtype=torch.uint8, device="cuda")
    assert L.fus_mass_gather_plan_build(None, N, 0, nd, ws0.data_ptr(), ws0.numel(), None) == 0
    assert L.fus_mass_apply_gather_f64(None, None, None, None, ws0.data_ptr(), N, 0, None) == 0
    L.fus_plan_release(ws0.data_ptr())


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("mode", [0, 2, 4], ids=["cached", "nontemporal-loads-stores", "nontemporal-stores"])
def test_vector_kernels_streaming_modes(gpu, dtype, mode):
    """FUS_TUNE_VECTOR_STREAM: the vector kernels with cached, non-temporal load + store and non-temporal store-only accesses
    (forced on, so that small vectors take the streaming paths too): aligned (16-byte accesses) and misaligned views, odd
    lengths and tails, every op; the fused RK4 stage kernels of both solvers, all four stage kinds, bit for bit equal to the
    cached mode (same arithmetic, only the access flavour differs)."""
    import torch

    dev, ops = gpu
    lib_mod = pkg("_lib")
    lib = lib_mod.load()
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    suf = "f64" if dtype == np.float64 else "f32"
    rng = np.random.default_rng(3)
    old = lib_mod.get_tuning(lib_mod.TUNE_VECTOR_STREAM)

    def run_all(n, off):
        a = torch.from_numpy(rng.standard_normal(n + off).astype(dtype)).cuda()[off:]
        b = torch.from_numpy((2 + rng.random(n + off)).astype(dtype)).cuda()[off:]
        res = {}
        y = b.clone()
        ops.axpy[1, 1](0.37, a, y)
        res["axpy"] = y.clone()
        out = torch.empty_like(a)
        ops.copy(a, out)
        res["copy"] = out.clone()
        ops.fill(1.5, out)
        res["fill"] = out.clone()
        ops.pointwise_divide[1, 1](a, b, out)
        res["divide"] = out.clone()
        ops.square[1, 1](a, out)
        res["square"] = out.clone()
        ops.scale(-2.5, a, out)
        res["scale"] = out.clone()
        yy = b.clone()
        lib_mod.check(getattr(lib, f"fus_muladd_{suf}")(a.data_ptr(), b.data_ptr(), yy.data_ptr(), n, lib_mod.stream_ptr()), "muladd")
        res["muladd"] = yy.clone()
        # fused stage kernels: 12 vectors, nlocal < ntotal (ghost block of b is zeroed, nothing else touched there)
        nl = n - min(n // 7, 5)
        for kind in (2, 0, 1, 3, 4, 5, 6, 7):  # 4 ... 7: the lean set of round 6
            vs = [torch.from_numpy(rng.standard_normal(n + off).astype(dtype)).cuda()[off:] for _ in range(8)]
            minv, bb, u, v, u0, v0, ku, un = vs
            minv.abs_().add_(1.0)
            lib_mod.check(getattr(lib, f"fus_rk4_stage_{suf}")(0.3, 0.7, kind, minv.data_ptr(), bb.data_ptr(), u.data_ptr(), v.data_ptr(), u0.data_ptr(),
                                                            v0.data_ptr(), ku.data_ptr(), un.data_ptr(), nl, n, lib_mod.stream_ptr()), "rk4_stage")
            res[f"rk4_{kind}"] = torch.cat([t.clone() for t in (bb, u, v, u0, v0, ku, un)])
            vs = [torch.from_numpy(rng.standard_normal(n + off).astype(dtype)).cuda()[off:] for _ in range(11)]
            m0, w2, w5, bb, u, v, u0, v0, ku, un, w = vs
            m0.abs_().add_(4.0)
            w2.mul_(0.01)
            lib_mod.check(getattr(lib, f"fus_rk4_stage_nl2_{suf}")(0.3, 0.7, kind, m0.data_ptr(), w2.data_ptr(), w5.data_ptr(), bb.data_ptr(), u.data_ptr(),
                                                                v.data_ptr(), u0.data_ptr(), v0.data_ptr(), ku.data_ptr(), un.data_ptr(), 0.25, w.data_ptr(), nl, n,
                                                                lib_mod.stream_ptr()), "rk4_stage_nl2")
            res[f"nl2_{kind}"] = torch.cat([t.clone() for t in (bb, u, v, u0, v0, ku, un, w)])
        torch.cuda.synchronize()
        return res

    try:
        for n in (1, 2, 3, 255, 257, 4099, 100003):
            for off in (0, 1):
                state = rng.bit_generator.state
                lib_mod.set_tuning(lib_mod.TUNE_VECTOR_STREAM, 0)
                ref = run_all(n, off)
                rng.bit_generator.state = state  # the same random inputs
                lib_mod.set_tuning(lib_mod.TUNE_VECTOR_STREAM, mode)
                got = run_all(n, off)
                for k in ref:
                    assert torch.equal(ref[k], got[k]), f"{k}, n = {n}, offset {off}, mode {mode}"
                # and the cached mode itself against numpy for the plain ops
                a = ref["copy"].cpu().numpy()
                assert np.allclose(ref["square"].cpu().numpy(), a * a, rtol=1e-6 if dtype == np.float32 else 1e-14)
    finally:
        lib_mod.set_tuning(lib_mod.TUNE_VECTOR_STREAM, old)
    assert lib_mod.get_tuning(lib_mod.TUNE_VECTOR_STREAM) == old
    with pytest.raises(lib_mod.FusGpuError):
        lib_mod.set_tuning(lib_mod.TUNE_VECTOR_STREAM, 9)


@pytest.mark.parametrize("westervelt", [False, True], ids=["linear", "westervelt"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
def test_lean_rk4_stage_kinds_equal_the_reference_sequence(gpu, dtype, westervelt):
    """``fus_rk4_stage_*`` / ``fus_rk4_stage_nl2_*`` through the C ABI: one step's four passes with the LEAN kinds 4, 5, 6, 7 (bw = dt / 6, aw = dt / 2;
    34 / 46 vector touches) against kinds 2, 0, 0, 3 (the reference's arithmetic operation for operation, cuda/demo_linear_box.py:491-563), fed the SAME
    right-hand sides b_1 ... b_4: the new u is BITWISE the same (its increments are formed with the same operations in the same order), the new v agrees to
    rounding (one re-derived term), the last stage's inputs (un, ku) are bitwise the same; ghost entries untouched, b re-zeroed; an unknown kind is refused."""
    import torch

    lib_mod = pkg("_lib")
    lib = lib_mod.load()
    suf = "f64" if dtype == np.float64 else "f32"
    rng = np.random.default_rng(11)
    n, nl = 10007, 9999
    dt = 0.37
    B, A = (1 / 6, 1 / 3, 1 / 3, 1 / 6), (0.0, 0.5, 0.5, 1.0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(dtype))).cuda()  # noqa: E731
    u_start, v_start = rng.standard_normal(n), rng.standard_normal(n)
    rhs = [rng.standard_normal(n) for _ in range(4)]
    minv = 1.0 + rng.random(n)
    m0, w2, w5 = 4.0 + rng.random(n), 0.01 * rng.standard_normal(n), rng.standard_normal(n)

    def one_step(kinds, coeff):
        u0, v0 = t(u_start), t(v_start)
        u, v, ku, un, b = (torch.full((n,), 7.0, dtype=u0.dtype, device="cuda") for _ in range(5))
        last_inputs = None
        for i in range(4):
            b.copy_(t(rhs[i]))
            bw, aw = coeff(i)
            if westervelt:
                rc = getattr(lib, f"fus_rk4_stage_nl2_{suf}")(bw, aw, kinds[i], t(m0).data_ptr(), t(w2).data_ptr(), t(w5).data_ptr(), b.data_ptr(), u.data_ptr(),
                                                           v.data_ptr(), u0.data_ptr(), v0.data_ptr(), ku.data_ptr(), un.data_ptr(), 0.0, None, nl, n, lib_mod.stream_ptr())
            else:
                rc = getattr(lib, f"fus_rk4_stage_{suf}")(bw, aw, kinds[i], t(minv).data_ptr(), b.data_ptr(), u.data_ptr(), v.data_ptr(), u0.data_ptr(), v0.data_ptr(),
                                                       ku.data_ptr(), un.data_ptr(), nl, n, lib_mod.stream_ptr())
            assert rc == 0
            torch.cuda.synchronize()
            assert float(b.abs().max()) == 0.0, "b is re-zeroed over owned and ghost entries"
            if i == 2:
                last_inputs = (un.clone(), ku.clone())
        return u0, v0, last_inputs

    ref = one_step((2, 0, 0, 3), lambda i: (B[i] * dt, 0.0 if i == 3 else A[i + 1] * dt))
    lean = one_step((4, 5, 6, 7), lambda i: (B[0] * dt, A[1] * dt))
    tol = 1e-14 if dtype == np.float64 else 1e-6
    if not westervelt:  # (the Westervelt pass feeds un and ku back into kv: its u differs with v's rounding from the second stage on)
        assert torch.equal(ref[0], lean[0]), "u: same operations in the same order"
        assert torch.equal(ref[2][0][:nl], lean[2][0][:nl])
    assert float((ref[0] - lean[0]).abs().max()) <= tol * float(ref[0].abs().max())
    assert float((ref[1] - lean[1]).abs().max()) <= tol * float(ref[1].abs().max())
    assert float((ref[2][1][:nl] - lean[2][1][:nl]).abs().max()) <= tol * float(ref[2][1].abs().max())
    for a_, b_ in ((ref[0], t(u_start)), (ref[1], t(v_start)), (lean[0], t(u_start)), (lean[1], t(v_start))):
        assert torch.equal(a_[nl:], b_[nl:]), "ghost entries of the solution are not touched"
    x = t(rhs[0])
    assert getattr(lib, f"fus_rk4_stage_{suf}")(0.1, 0.1, 8, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(),
                                            x.data_ptr(), nl, n, None) == -1
