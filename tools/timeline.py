#!/usr/bin/env python3
"""Print the tail of a rocprofv3 kernel trace as a timeline (start / end relative to the first shown dispatch, duration,
queue, kernel), to see which kernels of different streams really overlapped.

    python tools/timeline.py <dir with *_kernel_trace.csv> [last N dispatches, default 40] [skip last M]"""
import csv
import glob
import sys


def short(name):
    name = name.split("(")[0]
    for pre in ("void fus::", "fus::", "void "):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:60]


def main():
    files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) - n - skip: len(rows) - skip]
    t0 = int(rows[0]["Start_Timestamp"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"{s / 1e3:9.1f} {e / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  q{r.get('Queue_Id', '?'):>3} wg {int(r.get('Grid_Size', 0)) // max(int(r.get('Workgroup_Size', 1)), 1):6d}  {short(r['Kernel_Name'])}")


if __name__ == "__main__":
    main()
