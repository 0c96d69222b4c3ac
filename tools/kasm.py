#!/usr/bin/env python3
"""Extract one kernel's ISA from a -save-temps .s by (substring of) its mangled name and print an
instruction histogram + resource metadata.  usage: kasm.py file.s substring [--dump]"""
import collections, re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    m = re.match(r"^(\S+):\s*(;.*)?$", l)
    if m and key in m.group(1) and not m.group(1).startswith(".L") and start is None and "__hip" not in m.group(1):
        start, name = i, m.group(1)
    if start is not None and l.startswith(".Lfunc_end") :
        end = i
        break
body = lines[start:end]
hist = collections.Counter()
for l in body:
    m = re.match(r"^\s+([a-z][a-z0-9_]+)", l)
    if m:
        hist[m.group(1)] += 1
print(name, len(body), "lines")
for k, v in hist.most_common(60):
    print(f"{v:6d} {k}")
if "--dump" in sys.argv:
    print("\n".join(body))
