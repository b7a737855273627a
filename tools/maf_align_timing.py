#!/usr/bin/env python3
"""Wall time of pm_dp_align_maf (two MAF files of block pairs in, one MAF file of merged blocks out) on the GPU box:
python tools/maf_align_timing.py [pairs rows columns]; PM_TIMING=1 prints the library's own phase times if it has them."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from paramugsy_amd import dp  # noqa: E402

n, rows, L = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (2000, 8, 4096)
rng = np.random.default_rng(7)
tmp = tempfile.mkdtemp(prefix="mafalign")
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)


def write(path, tag, base=None):
    keep = []
    with open(path, "wb") as f:
        f.write(b"##maf version=1 scoring=test\n")
        for k in range(n):
            cons = alpha[rng.integers(0, 4, size=L)] if base is None else base[k].copy()
            if base is not None:
                m = rng.random(L) < 0.08
                cons[m] = alpha[rng.integers(0, 4, size=int(m.sum()))]
            keep.append(cons)
            f.write(b"a score=0 label=%d\n" % k)
            for r in range(rows):
                row = cons.copy()
                g = rng.random(L) < 0.05
                row[g] = ord("-")
                f.write(b"s %s.g%d %d %d + 100000000 " % (tag, r, 10 * k, int((~g).sum())))
                f.write(row.tobytes())
                f.write(b"\n")
            f.write(b"\n")
    return keep


pa, pb, po = os.path.join(tmp, "a.maf"), os.path.join(tmp, "b.maf"), os.path.join(tmp, "out.maf")
base = write(pa, b"L")
write(pb, b"R", base)
params = dp.make_params(rows, rows)
for rep in range(3):
    t = time.time()
    dp.align_maf_files(pa, pb, params, po)
    dt = time.time() - t
    cells = n * L * L
    print("pairs %d x %d rows x %d columns: %.3f s wall, in %.0f + %.0f MB, out %.0f MB, %.0f GCUPS end to end" %
          (n, rows, L, dt, os.path.getsize(pa) / 1e6, os.path.getsize(pb) / 1e6, os.path.getsize(po) / 1e6, cells / dt / 1e9), flush=True)
