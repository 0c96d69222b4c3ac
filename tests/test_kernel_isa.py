"""Pins the SHAPE of the compiled preambles (csrc/plan.hpp, "the preamble every planned kernel shares"): the gains of round 5's restructure
live in where the compiler puts its ``s_waitcnt`` -- a conditional table store, a sign extension inside a block, a predicated load whose
value is used in its own block each bring back one serial round trip (or a wait for the whole G slab) without changing any result.
hipcc cross-compiles gfx950 without a GPU; one device-only ``-S`` compile of four explicit instantiations (~30 s).

What is asserted is order, not exact instruction counts:
  * general-G kernel (P = 4, fp64, un-ordered plan, run tables): every load of the preamble -- table, run words, slots, the whole G slab
    -- is issued before the first vector wait, that wait leaves the slab in flight (vmcnt >= 15), and after the barrier of the run
    expansion the x gather is issued before any wait for "everything";
  * in-kernel-geometry kernel: the >= 12 loads of round trip 1 are issued before the first vector wait;
  * mass gather kernel: the row lengths, x and y go out together; in the first batch the entry ids are followed by ONE wait and then all
    eight detJ / constant loads;
  * PEER send kernel: all eight index loads of a thread are issued back to back."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "fenicsx-fus-gpu_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# The pins were taken with this compiler (``hipcc --version``: HIP 7.2.26015, AMD clang 22.0.0git roc-7.2.0).  Scheduling is the
# compiler's: another release may order the loads differently with no functional or performance regression, so on any other
# version a failing pin is reported as an expected failure (xfail, non-strict) instead of breaking the CPU suite.
PINNED_HIP_VERSION = "7.2.26015"


def _hip_version():
    try:
        out = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout
    except OSError:
        return None
    m = re.search(r"HIP version:\s*([0-9.]+)", out)
    return m.group(1) if m else None


_have_hipcc = shutil.which("hipcc") is not None or os.path.exists(HIPCC)
_pinned = _have_hipcc and _hip_version() == PINNED_HIP_VERSION
pytestmark = [pytest.mark.skipif(not _have_hipcc, reason="hipcc not available"),
              pytest.mark.xfail(condition=not _pinned, strict=False,
                                reason=f"ISA pins were taken with HIP {PINNED_HIP_VERSION}; this is {_hip_version()}: scheduling may differ")]

SOURCE = r"""
#include "stiffness_plan.hpp"
#include "stiffness_geom.hpp"
#include "mass_gather.hpp"
#include "halo_ipc.hpp"
namespace fus {
template __global__ void stiffness_plan_kernel<double, 4, 10, true, true, 1, 5, false, true>(const double*, const double*, double*, const double*, const int32_t*, const int32_t*, const uint16_t*, const double*, int64_t, int, const int32_t*, const int32_t*, LaunchSignal);
template __global__ void stiffness_plan_geom_kernel<double, 4, 10, true, true, 1, true, false, true>(const double*, const double*, double*, const double*, const int32_t*, const double*, const double*, const int32_t*, const int32_t*, const uint16_t*, const double*, int64_t, const int32_t*, const int32_t*, LaunchSignal);
template __global__ void mass_gather_kernel<double, 1, true, 2, false>(const double*, const double*, double*, const double*, GatherView, double, int, int64_t, GatherStatic);
template __global__ void ipc_send_kernel<double, true>(const double*, const int64_t*, int64_t, const IpcChunk*, const IpcPeer*, unsigned*, uint64_t*, uint64_t, uint64_t, IpcGate, int);
}
"""


@pytest.fixture(scope="module")
def kernels():
    """{mangled-name prefix: [instruction lines]} of the four kernels."""
    with tempfile.TemporaryDirectory() as d:
        src, out = os.path.join(d, "isa.hip"), os.path.join(d, "isa.s")
        with open(src, "w") as f:
            f.write(SOURCE)
        cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=fast", "-fno-slp-vectorize",
               "--cuda-device-only", "-S", "-I" + CSRC, "-o", out, src]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        text = open(out).read().split("\n")
    found = {}
    cur = None
    for ln in text:
        m = re.match(r"^(_ZN3fus\w+):", ln)
        if m:
            cur = m.group(1)
            found[cur] = []
            continue
        if cur is not None:
            s = ln.strip()
            if s and not s.startswith((";", ".")):
                found[cur].append(s)
            if s.startswith("s_endpgm"):
                cur = None
    return found


def _one(kernels, needle):
    hits = [v for k, v in kernels.items() if needle in k]
    assert len(hits) == 1, (needle, [k for k in kernels if needle in k])
    return hits[0]


def _vm_wait(s):
    m = re.match(r"s_waitcnt .*vmcnt\((\d+)\)", s)
    return int(m.group(1)) if m else None


def _is_load(s):
    return s.startswith(("global_load", "flat_load"))


def test_general_g_kernel_issues_everything_before_its_first_wait(kernels):
    k = _one(kernels, "21stiffness_plan_kernelIdLi4E")
    first_wait = next(i for i, s in enumerate(k) if _vm_wait(s) is not None)
    before = [s for s in k[:first_wait] if _is_load(s)]
    slab = [s for s in before if s.startswith("global_load_dwordx4")]
    assert len(slab) == 15, f"the whole G slab (5 planes x 3 x 16 bytes) must be in flight before the first wait, found {len(slab)}"
    assert len(before) >= 15 + 1 + 3 + 5, before  # + the dphi entry, the three run words, the five slots
    assert _vm_wait(k[first_wait]) >= 15, f"the first wait must leave the G slab in flight: {k[first_wait]}"
    # after the barrier of the run expansion: the x gather goes out before anything waits for every outstanding load
    barrier = next(i for i, s in enumerate(k) if s.startswith("s_barrier"))
    after = k[barrier:]
    gather = next(i for i, s in enumerate(after) if s.startswith("global_load_dwordx2"))
    assert all(_vm_wait(s) != 0 for s in after[:gather]), "a full wait before the x gather: the G slab would be waited for first"


def test_geometry_kernel_round_trip_one(kernels):
    k = _one(kernels, "26stiffness_plan_geom_kernelIdLi4E")
    first_wait = next(i for i, s in enumerate(k) if _vm_wait(s) is not None)
    before = [s for s in k[:first_wait] if _is_load(s)]
    # dphi entry, point, weight, three run words, one vertex id, five slots, the cell constant
    assert len(before) >= 12, before
    assert _vm_wait(k[first_wait]) >= 5, k[first_wait]


def test_mass_gather_kernel_batches(kernels):
    k = _one(kernels, "18mass_gather_kernelIdLi1ELb1ELi2ELb0E")
    first_wait = next(i for i, s in enumerate(k) if _vm_wait(s) is not None)
    before = [s for s in k[:first_wait] if _is_load(s)]
    assert sum(s.startswith("global_load_ubyte") for s in before) == 2 and sum(s.startswith("global_load_dwordx2") for s in before) == 4, \
        f"row lengths, x and y of both rows in one round trip: {before}"
    barrier = next(i for i, s in enumerate(k) if s.startswith("s_barrier"))
    after = k[barrier:]
    ids = [i for i, s in enumerate(after) if s.startswith("global_load_dword ")][:4]
    assert len(ids) == 4
    assert all(_vm_wait(s) is None for s in after[ids[0]:ids[3]]), "the four entry ids of the first batch go out without a wait between them"
    waits = [i for i, s in enumerate(after) if i > ids[3] and _vm_wait(s) is not None]
    dependents = [i for i, s in enumerate(after) if i > ids[3] and s.startswith("global_load_dwordx2")][:8]
    assert len(dependents) == 8
    # every wait between the entry ids and the last of the eight dependent loads comes BEFORE the first of them (the ids arriving)
    assert all(w < dependents[0] for w in waits if w < dependents[-1]), \
        "the detJ / constant loads of a batch must all be issued behind one wait for the entry ids"


def test_peer_send_kernel_indices_back_to_back(kernels):
    k = _one(kernels, "15ipc_send_kernelIdLb1E")
    barrier = next(i for i, s in enumerate(k) if s.startswith("s_barrier"))
    after = k[barrier:]
    idx = [i for i, s in enumerate(after) if s.startswith("global_load_dwordx2")][:8]
    assert len(idx) == 8
    assert all(_vm_wait(s) is None for s in after[idx[0]:idx[7]]), "the eight index loads of a thread are issued without a wait between them"
