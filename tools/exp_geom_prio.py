#!/usr/bin/env python3
"""Experiment (round 6): wave priority in the in-kernel-geometry kernel.  The kernel is bound by instruction issue and the four
waves of a SIMD belong to four workgroups in different phases: a wave in its preamble (index arithmetic + the loads everything else waits for)
competes for the issue port with waves in their arithmetic phases.  Builds of a COPY of csrc/ (tools/_exp/, never shipped):

  prio    s_setprio 3 from kernel entry until the x gather and the vertex coordinates have been issued, then 0
  prio2   the same, and s_setprio 1 again for the flush (its atomics are fire-and-forget: get them out, free the slot)

    python tools/exp_geom_prio.py build     # tools/_bin/libfusgpu_{prio,prio2}.so
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "fenicsx-fus-gpu_amd", "csrc")


def patch(text, variant):
    a = "  launch_signal_publish(sig);\n  constexpr int VPT = (CPB * 24 + BLOCK - 1) / BLOCK;"
    assert a in text
    text = text.replace(a, "  __builtin_amdgcn_s_setprio(3);  // EXPERIMENT\n" + a)
    b = "    stage_vertex_coords_issue<T, VPT, BLOCK, CPB>(x_g, vid, tid, cv);\n"
    assert b in text
    text = text.replace(b, b + "    __builtin_amdgcn_s_setprio(0);  // EXPERIMENT\n")
    if variant == "prio2":
        c = "  plan_flush<T, SPT, BLOCK>(y, mydof, nu_b, tid, sy);\n}\n\n// CPB: cells per batch"
        assert c in text
        text = text.replace(c, "  __builtin_amdgcn_s_setprio(1);  // EXPERIMENT\n" + c)
    return text


def build(variant):
    top = os.path.join(ROOT, "tools", "_exp", variant)
    dst = os.path.join(top, "fenicsx-fus-gpu_amd", "csrc")
    shutil.rmtree(top, ignore_errors=True)
    os.makedirs(os.path.dirname(dst))
    shutil.copytree(CSRC, dst, ignore=shutil.ignore_patterns("_obj", "_asm", "_ab", "*.so", "*.so.*", "fenicsx-fus-gpu_amd"))
    os.symlink(os.path.join(ROOT, "include"), os.path.join(top, "include"))
    p = os.path.join(dst, "stiffness_geom.hpp")
    with open(p) as f:
        t = f.read()
    with open(p, "w") as f:
        f.write(patch(t, variant))
    subprocess.run(["make", "-C", dst, "libfusgpu.so"], check=True, capture_output=True)
    out = os.path.join(ROOT, "tools", "_bin", f"libfusgpu_{variant}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    shutil.copy(os.path.join(dst, "libfusgpu.so"), out)
    print("built", out)


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        for v in ("prio", "prio2"):
            build(v)
    else:
        print(__doc__)
