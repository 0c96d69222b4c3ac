#!/usr/bin/env python3
"""Register / LDS / occupancy table of the kernels in libfusgpu.so, from hipcc's
``-Rpass-analysis=kernel-resource-usage`` remarks (no GPU needed: hipcc cross-compiles gfx950).

    python tools/resource_usage.py [regex ...]      # compile (if stale) and print matching kernels

``parse()`` / ``compile_remarks()`` are what tests/test_resource_usage.py pins the shipped builds with."""
import os
import re
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
CSRC = os.path.join(ROOT, "fenicsx-fus-gpu_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

_FIELDS = {
    "vgpr": r"VGPRs: (\d+)",
    "agpr": r"AGPRs: (\d+)",
    "sgpr": r"SGPRs: (\d+)",
    "scratch": r"ScratchSize \[bytes/lane\]: (\d+)",
    "occupancy": r"Occupancy \[waves/SIMD\]: (\d+)",
    "lds": r"LDS Size \[bytes/block\]: (\d+)",
}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else list(names)


def parse(text):
    """{demangled kernel name: {vgpr, agpr, sgpr, scratch, occupancy, lds}}"""
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    names, vals = [], []
    for b in blocks:
        names.append(b.split()[0])
        d = {}
        for k, pat in _FIELDS.items():
            m = re.search(pat, b)
            d[k] = int(m.group(1)) if m else None
        vals.append(d)
    return dict(zip(demangle(names), vals))


def compile_remarks(source="fus_gpu.hip", extra=()):
    """Device-only compile of ``source`` with resource-usage remarks; returns the remark text."""
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=fast",
           "-fno-slp-vectorize", "--cuda-device-only", "-c", "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *extra, source]
    r = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed:\n{r.stderr[-4000:]}")
    return r.stderr


def cached_remarks():
    """Remarks of the current sources, cached under csrc/_asm keyed on source mtimes."""
    cache = os.path.join(CSRC, "_asm", "resource_usage_device.txt")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(ROOT, "include", "fus_gpu.h"))
    newest = max(os.path.getmtime(p) for p in srcs)
    if os.path.exists(cache) and os.path.getmtime(cache) >= newest:
        return open(cache).read()
    text = compile_remarks()
    os.makedirs(os.path.dirname(cache), exist_ok=True)
    with open(cache, "w") as f:
        f.write(text)
    return text


def main():
    pats = [re.compile(p) for p in sys.argv[1:]] or [re.compile("stiffness_plan|westervelt_cell")]
    table = parse(cached_remarks())
    for name, d in table.items():
        if any(p.search(name) for p in pats):
            short = re.sub(r"\(.*", "", name).replace("void fus::", "")
            print(f"{short:80s} VGPR {d['vgpr']:4d} SGPR {d['sgpr']:4d} scratch {d['scratch']:4d} "
                  f"occ {d['occupancy']} LDS {d['lds']}")


if __name__ == "__main__":
    main()
