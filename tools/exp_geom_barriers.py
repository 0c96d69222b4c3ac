#!/usr/bin/env python3
"""VERDICT r5 item 3 -- what the cell-local barriers of the in-kernel-geometry kernel cost, as an UPPER BOUND: builds of a COPY of
csrc/ (tools/_exp/, never shipped) in which workgroup barriers are removed outright.  The kernels are then RACY (their results
are garbage; no index depends on an LDS value, so nothing can fault) -- only their time is meaningful: it is what an ideal
wave-local ordering of the same phases could reach at best.

  nobar23   B2 (after the u cube) and B3 (after the flux cubes) removed: the two barriers whose CELL-LOCAL duty a wave-aligned
            cell layout could take over (their batch-wide duties -- sx / sy alias the cubes -- would need LDS of their own)
  nobar     every barrier but the first (x values in LDS) removed: the bound of any rearrangement

    python tools/exp_geom_barriers.py build          # here (hipcc cross-compiles): tools/_bin/libfusgpu_{nobar23,nobar}.so
    bash profiles/scripts/r06f.sh                    # on the GPU box: shipped | nobar23 | nobar, P = 3, 4, 7, interleaved
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "fenicsx-fus-gpu_amd", "csrc")


def patch_geom(text, variant):
    b2 = """      cu[ix * n2] = u[ix];
    }
  }
  __syncthreads();
  if constexpr (!ALIAS) plan_zero"""
    assert b2 in text
    text = text.replace(b2, b2.replace("  __syncthreads();\n", "  __builtin_amdgcn_wave_barrier();  // EXPERIMENT: B2 removed\n"))
    b3 = """      cfz[qx * n2] = gq[2] * vx + gq[4] * vy + gq[5] * vz;
    }
  }
  __syncthreads();
  if constexpr (ALIAS) {"""
    assert b3 in text
    text = text.replace(b3, b3.replace("  __syncthreads();\n", "  __builtin_amdgcn_wave_barrier();  // EXPERIMENT: B3 removed\n"))
    if variant == "nobar":
        b3p = """    plan_zero<T, SPT, BLOCK>(sy, nu_b, tid);
    __syncthreads();
  }

  plan_backward"""
        assert b3p in text
        text = text.replace(b3p, b3p.replace("    __syncthreads();\n", "    __builtin_amdgcn_wave_barrier();  // EXPERIMENT: B3' removed\n"))
    return text


def patch_plan_header(text, variant):
    if variant != "nobar":
        return text
    # plan_backward's trailing barrier (before the flush) and the run expansion's
    b4 = """      lds_atomic_add(&sy[sl[jx]], (PlanAcc)acc);
    }
  }
  __syncthreads();
}"""
    assert b4 in text
    return text.replace(b4, b4.replace("  __syncthreads();\n", "  __builtin_amdgcn_wave_barrier();  // EXPERIMENT: B4 removed\n"))


def build(variant):
    top = os.path.join(ROOT, "tools", "_exp", variant)
    dst = os.path.join(top, "fenicsx-fus-gpu_amd", "csrc")
    shutil.rmtree(top, ignore_errors=True)
    os.makedirs(os.path.dirname(dst))
    shutil.copytree(CSRC, dst, ignore=shutil.ignore_patterns("_obj", "_asm", "_ab", "*.so", "*.so.*", "fenicsx-fus-gpu_amd"))
    os.symlink(os.path.join(ROOT, "include"), os.path.join(top, "include"))
    for name, fn in (("stiffness_geom.hpp", patch_geom), ("stiffness_plan.hpp", patch_plan_header)):
        p = os.path.join(dst, name)
        with open(p) as f:
            t = f.read()
        with open(p, "w") as f:
            f.write(fn(t, variant))
    subprocess.run(["make", "-C", dst, "libfusgpu.so"], check=True, capture_output=True)
    out = os.path.join(ROOT, "tools", "_bin", f"libfusgpu_{variant}.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    shutil.copy(os.path.join(dst, "libfusgpu.so"), out)
    print("built", out)


if __name__ == "__main__":
    if sys.argv[1:] == ["build"]:
        for v in ("nobar23", "nobar"):
            build(v)
    else:
        print(__doc__)
