"""What every part of the bench shares: the repo root, the roofline's denominator, stderr logging, the ONE JSON line on
stdout, the watchdog, host-core count, hashes that tie a line to the library / kernel sources it ran."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


_JSON_FD = None


def protect_stdout():
    """The contract is ONE JSON line on stdout.  Native libraries loaded below write there too (RCCL
    prints a version banner on stdout when a communicator is created with ncclCommInitRank), so file
    descriptor 1 is pointed at stderr for the life of the process and the JSON line is written to a
    private duplicate of the original stdout."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def start_watchdog(seconds, rank):
    """A rank that is still running after ``seconds`` is taken to be hung (a collective whose peer never
    arrived, a kernel that never drains): say so and leave with a non-zero code, so that the launcher
    stops the other ranks and the caller sees a failure instead of a job that never ends."""
    import threading

    if seconds <= 0:
        return

    def fire():
        log(f"rank {rank}: watchdog: still running after {seconds:.0f} s -- giving up (FUS_BENCH_WATCHDOG_S=0 disables)")
        os._exit(124)

    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def lib_sha():
    """Short hash of the libfusgpu.so this run loads (ties a bench line to the profiled binary)."""
    import hashlib

    import fusgpu_loader

    path = fusgpu_loader.submodule("_lib").LIB_PATH
    try:
        with open(path, "rb") as f:
            return hashlib.sha256(f.read()).hexdigest()[:12]
    except OSError:
        return None


def kernel_src_sha(files=("plan.hpp", "stiffness.hpp", "stiffness_plan.hpp")):
    """Hash of what defines a kernel's code (its sources and the compile flags; default: the HEADLINE kernel): a PMC pass
    stays valid for a library that differs from the profiled one only elsewhere (halo transport, ABI glue)."""
    import hashlib

    csrc = os.path.join(ROOT, "fenicsx-fus-gpu_amd", "csrc")
    h = hashlib.sha256()
    try:
        for n in ("Makefile",) + tuple(files):
            with open(os.path.join(csrc, n), "rb") as f:
                data = f.read()
            if n == "Makefile":  # only the compile flags: the header list changes with every new file
                data = b"\n".join(line for line in data.split(b"\n") if line.startswith((b"CXXFLAGS", b"ARCH", b"           -f")))
            h.update(data)
    except OSError:
        return None
    return h.hexdigest()[:12]


def lib_built_from_tree():
    """True if the loaded libfusgpu.so was built from exactly the sources in this tree (fus_source_hash())."""
    import fusgpu_loader

    try:
        return bool(fusgpu_loader.submodule("_lib").built_from_tree())
    except Exception:
        return None


def _free_port():
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def rehearsal():
    """FUS_BENCH_REHEARSAL=1: the N-rank code path of THIS script with real HIP kernels and real processes
    where only one GPU exists -- every rank on the visible GPU(s) modulo their count, torch.distributed over
    gloo, the exchange staged through the host.  The line is marked invalid: not a measurement."""
    return os.environ.get("FUS_BENCH_REHEARSAL", "0") == "1"


def coll_device(device):
    """Where the tensors of bench.py's own collectives (barrier flags, max-over-ranks time) live."""
    import torch

    return torch.device("cpu") if rehearsal() else device
