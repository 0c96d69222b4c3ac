#!/usr/bin/env python3
"""Why did tests/test_halo_gpu.py::test_fork_join_enforce_one_caller_stream once read 2 % of a 2 GiB vector one add behind
AFTER a device synchronise?  Repeats the scenario in variants to see which ingredient it needs:

  A  torch only: adds on a high-priority torch stream, device synchronise, count on the default stream
  B  the same on the communicator's stream (an ExternalStream over the library's high-priority non-blocking stream)
  C  B with the library's fork / join around the add (the test's phase 2)
  D  the whole test sequence (40 adds behind a fork from s_a, refused join / fork from s_b, join, then phase 2)
  D1-D4  D with none / some of the refused calls;  D5  D with a device synchronise between ``torch.zeros`` and the first use

Finding (profiles/r05c_stream_visibility_probe.log): A, B, C never fail; D and D1-D4 fail in about half of the repetitions whatever
sits in the middle -- the vector is allocated and ZERO-FILLED ON THE DEFAULT STREAM and first used on non-blocking streams with
no ordering in between, so the first adds overtake the fill on part of the vector.  A test bug, not a library one; D5 is the fix.

Each variant is repeated ``--reps`` times; prints the number of repetitions with a wrong count and the worst count."""
import argparse
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--log2", type=int, default=28)
    a = ap.parse_args()
    import torch

    import fusgpu_loader

    scat, lib_mod = fusgpu_loader.submodule("scatterer"), fusgpu_loader.submodule("_lib")
    torch.cuda.set_device(0)
    n = 1 << a.log2

    def count(big, want):
        return int((big != want).sum().item())

    def variant_a():
        big = torch.zeros(n, dtype=torch.float64, device="cuda")
        hp = torch.cuda.Stream(priority=-1)
        torch.cuda.synchronize()
        with torch.cuda.stream(hp):
            for _ in range(41):
                big.add_(1.0)
        hp.synchronize()
        torch.cuda.synchronize()
        return count(big, 41.0)

    def variant_b(comm):
        big = torch.zeros(n, dtype=torch.float64, device="cuda")
        side = comm.stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(41):
                big.add_(1.0)
        side.synchronize()
        torch.cuda.synchronize()
        return count(big, 41.0)

    def variant_c(comm):
        big = torch.zeros(n, dtype=torch.float64, device="cuda")
        side = comm.stream()
        s = torch.cuda.Stream()
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            for k in range(3):
                comm.fork()
                with torch.cuda.stream(side):
                    big.add_(1.0)
                comm.join()
        side.synchronize()
        torch.cuda.synchronize()
        return count(big, 3.0)

    def variant_d(comm, middle="both", sync_after_fill=False):
        big = torch.zeros(n, dtype=torch.float64, device="cuda")  # the fill runs on the DEFAULT stream; every stream below is non-blocking
        if sync_after_fill:
            torch.cuda.synchronize()
        side = comm.stream()
        s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(s_a):
            comm.fork()
            with torch.cuda.stream(side):
                for _ in range(40):
                    big.add_(1.0)
        with torch.cuda.stream(s_b):
            if middle == "query":
                side.query()  # hipStreamQuery on the busy communicator stream, nothing else
            for fn in {"both": (comm.join, comm.fork), "join": (comm.join,), "fork": (comm.fork,)}.get(middle, ()):
                try:
                    fn()
                    raise SystemExit("misuse accepted")
                except lib_mod.FusGpuError:
                    pass
        with torch.cuda.stream(s_a):
            comm.join()
        torch.cuda.synchronize()
        first = count(big, 40.0)
        with torch.cuda.stream(s_b):
            comm.fork()
            with torch.cuda.stream(side):
                big.add_(1.0)
            comm.join()
        side.synchronize()
        torch.cuda.synchronize()
        return first * 1000000007 + count(big, 41.0) if first else count(big, 41.0)

    comm = scat.NativeComm(transport="peer")
    for name, fn in (("A torch high-priority stream", variant_a), ("B communicator stream", lambda: variant_b(comm)),
                     ("C fork / add / join x 3", lambda: variant_c(comm)), ("D the test's sequence", lambda: variant_d(comm)),
                     ("D1 no refused calls", lambda: variant_d(comm, "none")), ("D2 hipStreamQuery only", lambda: variant_d(comm, "query")),
                     ("D3 refused join only", lambda: variant_d(comm, "join")), ("D4 refused fork only", lambda: variant_d(comm, "fork")),
                     ("D5 = D, synchronise after the fill", lambda: variant_d(comm, "both", True))):
        bad = []
        for _ in range(a.reps):
            w = fn()
            if w:
                bad.append(w)
        print(f"{name:32s}: {len(bad)} of {a.reps} repetitions wrong" + (f" (worst {max(bad)} of {n} elements)" if bad else ""), flush=True)
    print("health", comm.health())
    comm.close()


if __name__ == "__main__":
    main()
