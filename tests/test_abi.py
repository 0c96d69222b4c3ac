"""The C-ABI library loads and exports every symbol include/fus_gpu.h declares;
argument validation happens before any device work (no compute calls here)."""

import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, pkg


def declared_functions():
    hdr = open(os.path.join(ROOT, "include", "fus_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(fus_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    lib_mod = pkg("_lib")
    assert os.path.exists(lib_mod.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    raw = C.CDLL(lib_mod.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 30
    for name in names:
        assert hasattr(raw, name), f"{name} declared in include/fus_gpu.h but not exported"
    # and the ctypes binding covers the same set
    bound = set(lib_mod.SIGNATURES) | {"fus_error_string", "fus_source_hash"}
    assert bound == set(names), sorted(bound ^ set(names))


def test_host_only_entry_points():
    lib_mod = pkg("_lib")
    lib = lib_mod.load()
    assert lib.fus_abi_version() == lib_mod.ABI_VERSION == 3  # include/fus_gpu.h FUS_ABI_VERSION
    assert lib.fus_error_string(0) == b"ok"
    assert b"degree" in lib.fus_error_string(-2)
    assert lib.fus_stiffness_plan_bytes(4, 157464) > 0
    assert lib.fus_stiffness_plan_bytes(11, 10) == -2
    old = lib_mod.get_tuning(lib_mod.TUNE_STIFFNESS_VARIANT)
    lib_mod.set_tuning(lib_mod.TUNE_STIFFNESS_VARIANT, 1)
    assert lib_mod.get_tuning(lib_mod.TUNE_STIFFNESS_VARIANT) == 1
    lib_mod.set_tuning(lib_mod.TUNE_STIFFNESS_VARIANT, old)
    assert lib.fus_set_tuning(999, 0) == -1


def test_argument_validation_precedes_device_work():
    lib = pkg("_lib").load()
    z = C.c_void_p(0)
    one = C.c_void_p(256)  # non-null, never dereferenced: validation fails first
    assert lib.fus_stiffness_apply_f64(one, one, one, one, one, one, 11, 1, z) == -2  # unsupported degree
    assert lib.fus_stiffness_apply_f64(one, one, one, one, one, one, 0, 1, z) == -2
    assert lib.fus_stiffness_apply_f64(z, one, one, one, one, one, 4, 1, z) == -1  # null x
    assert lib.fus_stiffness_apply_f64(one, one, one, one, one, one, 4, -1, z) == -1  # negative size
    assert lib.fus_stiffness_apply_f64(one, one, one, C.c_void_p(264), one, one, 4, 1, z) == -1  # G not 16-B aligned
    assert lib.fus_stiffness_apply_f64(z, z, z, z, z, z, 4, 0, z) == 0  # zero cells: no-op
    assert lib.fus_mass_apply_f64(z, z, z, z, z, 125, 0, z) == 0
    assert lib.fus_mass_apply_f64(one, one, one, one, one, 0, 1, z) == -1
    assert lib.fus_axpy_f64(1.0, z, z, 0, z) == 0
    assert lib.fus_axpy_f64(1.0, z, one, 5, z) == -1
    assert lib.fus_pack_fwd_f64(z, z, z, 0, z) == 0
    assert lib.fus_unpack_rev_f64(one, z, one, 3, z) == -1
    assert lib.fus_stiffness_plan_build(one, 4, 10, C.c_void_p(257), 1 << 30, z) == -1  # misaligned workspace
    assert lib.fus_stiffness_plan_build(one, 4, 10, one, 16, z) == -1  # workspace too small


def test_product_has_no_cpu_fallback():
    """Host tensors must be rejected loudly, never routed to a CPU path."""
    import torch

    ops, lib_mod = pkg("operators"), pkg("_lib")
    t = torch.zeros(8, dtype=torch.float64)
    with pytest.raises(lib_mod.FusGpuError):
        ops.fill(1.0, t)
    with pytest.raises(lib_mod.FusGpuError):
        ops.mass_operator(8, np.float64)(t, t[:1], t, t.reshape(1, 8), torch.zeros(1, 8, dtype=torch.int32))
    with pytest.raises(TypeError):
        ops.copy(np.zeros(4), np.zeros(4))  # numpy arrays are not device arrays


def test_product_does_not_import_oracle():
    pkgdir = os.path.join(ROOT, "fenicsx-fus-gpu_amd")
    for fn in os.listdir(pkgdir):
        if fn.endswith(".py"):
            src = open(os.path.join(pkgdir, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
    # bench.py's implementation (benchlib/) may use the oracle as CPU baseline and as checker only: the modules that hold those legs import it,
    # the modules that launch and time kernels (steps, aux_lines, harvest, roofline, launch, common) reach it through cpu_legs alone
    allowed = {"cpu_legs.py", "transports.py", "apply.py"}  # the CPU legs; the scatter line's CPU leg; the N > 1 checker's portable-library build
    for fn in os.listdir(os.path.join(ROOT, "benchlib")):
        if fn.endswith(".py") and fn not in allowed:
            src = open(os.path.join(ROOT, "benchlib", fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"benchlib/{fn}"


@pytest.mark.gpu
def test_c_abi_demo_plain_cpp_host():
    """examples/c_abi_demo.cpp: a C++ host with no Python/torch drives the library through the C ABI
    alone and checks operator identities (K 1 = 0, symmetry, exact energy, planned == plan-free)."""
    import subprocess

    exe = os.path.join(ROOT, "examples", "c_abi_demo")
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)  # rebuilds if a source or the header is newer
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "C_ABI_DEMO_OK" in r.stdout, r.stdout + r.stderr


def test_raw_golden_twin_matches_npz():
    """tests/golden/ops_P4_2x2x2_pert_float64.bin (read by examples/c_abi_golden.cpp) is the same data as
    the reference-generated .npz."""
    import sys

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import export_raw

    d = np.load(os.path.join(ROOT, "tests", "golden", export_raw.CASE + ".npz"))
    with open(os.path.join(ROOT, "tests", "golden", export_raw.CASE + ".bin"), "rb") as f:
        assert f.read() == export_raw.encode(d)


@pytest.mark.gpu
def test_c_abi_golden_plain_cpp_host():
    """examples/c_abi_golden.cpp: a C++ host with no Python/torch compares stiffness (three entry points),
    cell / facet mass and the device precompute with the REFERENCE'S outputs (golden twin), and runs a
    fus_halo_* exchange in a 1-rank world over the RCCL transport and over the PEER transport (arena blob export ->
    connect, exchanges on the communicator's stream between fus_comm_fork and fus_comm_join)."""
    import subprocess

    exe = os.path.join(ROOT, "examples", "c_abi_golden")
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)  # rebuilds if the source is newer
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "ops_P4_2x2x2_pert_float64.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "C_ABI_GOLDEN_OK" in r.stdout, r.stdout + r.stderr
    assert "halo reverse (PEER transport, own arena)" in r.stdout and "halo forward (RCCL send/recv to self)" in r.stdout
    # the C++ functor twins of cpp/common/spectral_op.hpp (include/fus_gpu.hpp) against the reference's outputs
    assert "StiffnessSpectral3D<double,4>::operator()" in r.stdout and "MassSpectral3D<double,4>::operator()" in r.stdout
    assert "atomic-free kernel, detJ in row order" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("geometry,warp", [(0, 0), (1, 0), (1, 1), (2, 1)], ids=["affine", "generalG", "generalG-warped", "in-kernel-warped"])
def test_c_abi_linear_box_plain_cpp_host(tmp_path, geometry, warp):
    """examples/c_abi_linear_box.cpp: the demo_linear_box RK4 loop as a C++ host over the C ABI alone (class shape
    of cpp/common/Linear.hpp), on a box it builds itself (GLL tables, mesh, facets, time-step rule in C++).
    Same pressure field as the Python driver (linear_solver.LinearSpectral3D) on the same box."""
    import subprocess
    import sys

    import numpy as np

    sys.path.insert(0, ROOT)
    import fusgpu_loader

    exe = os.path.join(ROOT, "examples", "c_abi_linear_box")
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, capture_output=True)
    P, N, steps, L = 4, 6, 30, 0.12
    out = str(tmp_path / "u.bin")
    r = subprocess.run([exe, str(P), str(N), str(steps), str(geometry), str(warp), out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    u_cpp = np.fromfile(out, dtype=np.float64)

    boxmesh, ls = fusgpu_loader.submodule("boxmesh"), fusgpu_loader.submodule("linear_solver")
    h = L / N

    def warp_fn(xg):  # the displacement examples/c_abi_linear_box.cpp applies
        o = xg.copy()
        s = np.sin(np.pi * xg[:, 0] / L)
        y, z = xg[:, 1] / L, xg[:, 2] / L
        o[:, 0] += 0.15 * h * s * np.sin(2 * np.pi * y) * np.cos(2 * np.pi * z)
        o[:, 1] += 0.10 * h * s * np.cos(2 * np.pi * z)
        o[:, 2] += 0.10 * h * s * np.sin(2 * np.pi * y)
        return o

    mesh = boxmesh.BoxMesh(P, N, length=L, warp=warp_fn if warp else None)
    solver = ls.LinearSpectral3D(mesh, np.float64, in_kernel_geometry=(geometry == 2))
    assert solver.affine == (warp == 0)
    hmin = ls.time_step_parameters(mesh, P, 1500.0, 0.5e6, L)
    dt, tf, _ = ls.snap_time_step(hmin, P, 1500.0, 0.5e6, L)
    dt_cpp = float(r.stdout.split("dt=")[1].split()[0])
    assert abs(dt_cpp - dt) <= 1e-14 * dt, (dt_cpp, dt)
    solver.init()
    _, done = solver.rk4(0.0, tf, dt, max_steps=steps)
    assert done == steps and f"steps={steps} " in r.stdout
    u_py = solver.u_sol()
    assert u_cpp.shape == u_py.shape
    scale = np.max(np.abs(u_py))
    assert scale > 0 and np.max(np.abs(u_cpp - u_py)) < 1e-10 * scale, np.max(np.abs(u_cpp - u_py)) / scale


def test_committed_pmc_passes_belong_to_the_kernels_in_this_tree():
    """bench.py replays HBM traffic from profiles/traffic_latest.json only for a library built from the kernel sources and
    compile flags that were profiled.  This test fails when a kernel source changed after the last PMC pass: re-profile
    (profiles/run_profile.sh + summarize.py) instead of shipping a line whose ``roofline.traffic`` would be null."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench

    t = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    assert t["kernel_src_sha"] == bench.kernel_src_sha(), "stiffness kernel sources changed since profiles/" + t["source"]
    tm = t["aux"]["mass"]
    assert tm["kernel_src_sha"] == bench.kernel_src_sha(tuple(tm.get("kernel_src_files", ("plan.hpp", "mass.hpp")))), "mass kernel sources changed"
    assert tm.get("kernel") == "fus::mass_gather_kernel"  # what aux.mass of the default line launches at config 3
    assert os.path.exists(os.path.join(ROOT, t["source"])) and os.path.exists(os.path.join(ROOT, t["aux"]["mass"]["source"]))
    # the other replayed lines of the default bench line: in-kernel geometry, both RK4 steps, the Westervelt steps
    for key in ("stiffness_in_kernel_geometry", "rk4_step", "rk4_step_in_kernel_geometry", "westervelt_step", "westervelt_step_in_kernel_geometry",
                "westervelt_step_in_kernel_geometry_single_gather"):
        e = t["aux"][key]
        assert e["kernel_src_sha"] == bench.kernel_src_sha(tuple(e["kernel_src_files"])), f"{key}: kernel sources changed since {e['source']}"
        assert os.path.exists(os.path.join(ROOT, e["source"])), e["source"]
