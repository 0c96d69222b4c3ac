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


# the translation units of libfusgpu.so (csrc/Makefile): (source, defines)
UNITS = [("fus_gpu.hip", ())] + [(f"{d}.hip", (f"-DFUS_INST_T={t}",)) for d in ("dispatch_stiffness_plan", "dispatch_geometry", "dispatch_westervelt")
                                   for t in ("double", "float")]


def compile_remarks(source=None, extra=()):
    """Device-only compile with resource-usage remarks of ``source`` (default: every translation unit of the library, in parallel);
    returns the remark text."""
    units = UNITS if source is None else [(source, ())]
    procs = []
    for src, defs in units:
        cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-ffp-contract=fast", "-fno-slp-vectorize",
               "--cuda-device-only", "-c", "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *defs, *extra, src]
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    text = ""
    for cmd, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"{' '.join(cmd)} failed:\n{err[-4000:]}")
        text += err
    return text


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
