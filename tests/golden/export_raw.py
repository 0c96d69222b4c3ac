#!/usr/bin/env python3
"""Raw-binary twin of one golden case for hosts without numpy (examples/c_abi_golden.cpp):

    python tests/golden/export_raw.py            # rewrites tests/golden/ops_P4_2x2x2_pert_float64.bin

The .npz it is made from was produced by the reference itself (generate_golden.py); this script only
re-encodes those arrays.  Layout: 8-byte magic "FUSGOLD1", int64 entry count, then per entry
name[32] (NUL padded), int32 dtype (0 = float64, 1 = int32, 2 = int64), int32 pad, int64 element count,
raw little-endian data (padded to 8 bytes).  tests/test_abi.py checks the twin against the .npz."""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASE = "ops_P4_2x2x2_pert_float64"
FIELDS = ["P", "x", "cell_constants", "y0", "dofmap", "dphi_1d", "pts", "wts", "x_dofs", "x_g", "ref_G", "ref_detJ",
          "ref_y_stiffness", "ref_y_mass", "facet_constants", "ref_detJ_f", "bfacet_dofmap", "ref_y_facet_mass"]
CODES = {np.dtype(np.float64): 0, np.dtype(np.int32): 1, np.dtype(np.int64): 2}


def encode(d):
    out = [b"FUSGOLD1", struct.pack("<q", len(FIELDS))]
    for name in FIELDS:
        a = np.ascontiguousarray(d[name])
        if a.dtype not in CODES:
            a = a.astype(np.int64)
        raw = a.tobytes()
        out.append(name.encode().ljust(32, b"\0"))
        out.append(struct.pack("<iiq", CODES[a.dtype], 0, a.size))
        out.append(raw + b"\0" * (-len(raw) % 8))
    return b"".join(out)


if __name__ == "__main__":
    d = np.load(os.path.join(HERE, CASE + ".npz"))
    blob = encode(d)
    with open(os.path.join(HERE, CASE + ".bin"), "wb") as f:
        f.write(blob)
    print("wrote", CASE + ".bin", len(blob), "bytes")
