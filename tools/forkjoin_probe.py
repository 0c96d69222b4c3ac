#!/usr/bin/env python3
"""What does the fork / join around the concurrent schedule cost by itself?

K back-to-back config-3 stiffness applies on the main stream, with per apply
  a  nothing else
  b  fork + join through two events and an EMPTY side stream           (torch events: hipEventDisableTiming)
  c  the same with one tiny kernel on the side stream
  d  b with events created as hipEventDisableTiming | hipEventReleaseToDevice (no system-scope release at the record)
  e  join only (side chain ordered after the previous apply's chain, as in a loop with nothing between applies)
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch

    import fusgpu_loader
    from conftest import build_problem

    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    ops, _lib = fusgpu_loader.submodule("operators"), fusgpu_loader.submodule("_lib")
    pb = build_problem(4, 54, perturb=0.16)
    mesh = pb["mesh"]
    x, cc, G = (torch.from_numpy(pb[k]).to(dev) for k in ("x", "cc", "G"))
    dm = torch.from_numpy(mesh.dofmap).to(dev)
    y = torch.zeros(mesh.ndofs, dtype=torch.float64, device=dev)
    z = torch.zeros(4096, dtype=torch.float64, device=dev)
    op = ops.stiffness_operator(4, pb["D"].flatten(), np.float64)
    op.prepare(dm)
    lib = _lib.load()
    hip = ctypes.CDLL("libamdhip64.so")
    main_s = torch.cuda.current_stream()
    side = torch.cuda.Stream(priority=-1)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    def apply():
        op(x, cc, y, G, dm)

    ea, eb = torch.cuda.Event(), torch.cuda.Event()

    def forkjoin(tiny=False, fork=True):
        if fork:
            ea.record(main_s)
            side.wait_event(ea)
        apply()
        if tiny:
            with torch.cuda.stream(side):
                lib.fus_fill_f64(ctypes.c_double(1.0), z.data_ptr(), 4096, _lib.stream_ptr())
        eb.record(side)
        main_s.wait_event(eb)

    # raw HIP events with hipEventReleaseToDevice
    DISABLE_TIMING, RELEASE_TO_DEVICE = 0x2, 0x40000000
    raw = []
    for _ in range(2):
        ev = ctypes.c_void_p()
        rc = hip.hipEventCreateWithFlags(ctypes.byref(ev), ctypes.c_uint(DISABLE_TIMING | RELEASE_TO_DEVICE))
        assert rc == 0, rc
        raw.append(ev)
    ms, ss = ctypes.c_void_p(main_s.cuda_stream), ctypes.c_void_p(side.cuda_stream)

    def forkjoin_raw(tiny=False):
        hip.hipEventRecord(raw[0], ms)
        hip.hipStreamWaitEvent(ss, raw[0], 0)
        apply()
        if tiny:
            with torch.cuda.stream(side):
                lib.fus_fill_f64(ctypes.c_double(1.0), z.data_ptr(), 4096, _lib.stream_ptr())
        hip.hipEventRecord(raw[1], ss)
        hip.hipStreamWaitEvent(ms, raw[1], 0)

    def side_kernels(k, n):
        def f():
            ea.record(main_s)
            side.wait_event(ea)
            apply()
            with torch.cuda.stream(side):
                for _ in range(k):
                    lib.fus_fill_f64(ctypes.c_double(1.0), zz.data_ptr(), n, _lib.stream_ptr())
            eb.record(side)
            main_s.wait_event(eb)
        return f

    zz = torch.zeros(1 << 20, dtype=torch.float64, device=dev)
    # how much does each small kernel on the high-priority side stream cost the chip-filling launch next to it?
    for n in (4096, 1 << 17):
        base = []
        rows = {k: [] for k in (0, 1, 2, 4, 8)}
        for _ in range(5):
            base.append(timed(apply))
            for k in rows:
                rows[k].append(timed(side_kernels(k, n)))
        b = float(np.median(base))
        print(f"side kernels of {n} elements ({n * 8 // 1024} KiB filled each): plain {b:7.1f} us | " +
              " | ".join(f"{k} kernels {float(np.median(v)):7.1f} ({float(np.median(np.array(v) - np.array(base))):+5.1f})" for k, v in rows.items()), flush=True)

    # what would a fork WITHOUT an event cost the main stream?  f: a one-element kernel in front of every apply;
    # g: hipStreamWriteValue64 in front of every apply (both followed by nothing on the side stream)
    flag = torch.zeros(8, dtype=torch.int64, device=dev)

    def tiny_then_apply():
        lib.fus_fill_f64(ctypes.c_double(1.0), z.data_ptr(), 1, _lib.stream_ptr())
        apply()

    seqc = [0]

    def writevalue_then_apply():
        seqc[0] += 1
        hip.hipStreamWriteValue64(ms, ctypes.c_void_p(flag.data_ptr()), ctypes.c_uint64(seqc[0]), 0)
        apply()

    rows = {"plain": [], "f tiny kernel before each apply": [], "g hipStreamWriteValue64 before each apply": [], "b fork+join events": []}
    for _ in range(5):
        rows["plain"].append(timed(apply))
        rows["f tiny kernel before each apply"].append(timed(tiny_then_apply))
        rows["g hipStreamWriteValue64 before each apply"].append(timed(writevalue_then_apply))
        rows["b fork+join events"].append(timed(forkjoin))
    base = np.array(rows["plain"])
    print("event-free fork candidates: " + " | ".join(f"{k} {float(np.median(v)):7.1f} ({float(np.median(np.array(v) - base)):+5.1f})" for k, v in rows.items()), flush=True)

    for rnd in range(1):
        ta = timed(apply)
        tb = timed(forkjoin)
        tc = timed(lambda: forkjoin(True))
        td = timed(forkjoin_raw)
        td2 = timed(lambda: forkjoin_raw(True))
        te = timed(lambda: forkjoin(True, fork=False))
        print(f"round {rnd}: a plain {ta:7.1f} us | b fork+join, empty side {tb:7.1f} ({tb - ta:+5.1f}) | c + tiny side kernel {tc:7.1f} ({tc - ta:+5.1f}) | "
              f"d ReleaseToDevice events {td:7.1f} ({td - ta:+5.1f}), + tiny kernel {td2:7.1f} ({td2 - ta:+5.1f}) | e join only + tiny kernel {te:7.1f} ({te - ta:+5.1f})",
              flush=True)


if __name__ == "__main__":
    main()
