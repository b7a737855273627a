#!/usr/bin/env python3
"""GPU box: PM_TIMING=1 python tools/stream_timing.py -- the phases of pm_dp_stream_align / pm_dp_stream_align_text on BASELINE
configs[1]'s shape (10 000 pairs of 2 rows x 1 000 columns) from pinned host buffers."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from paramugsy_amd import dp  # noqa: E402

n, rows, L = 10000, 2, 1000
inputs, side_a, side_b = dp.synth_pairs_fast(20261003, n, rows, L, with_rows=True)
params = dp.make_params(rows, rows)
pa, pb = dp.PinnedArray(inputs.cols_a.shape, np.uint8), dp.PinnedArray(inputs.cols_b.shape, np.uint8)
pa.a[...] = inputs.cols_a
pb.a[...] = inputs.cols_b
pin = dp.DpInputs(pa.a, inputs.off_a, pb.a, inputs.off_b)
ps, pn = dp.PinnedArray((n,), np.int32), dp.PinnedArray((n,), np.int32)
po = dp.PinnedArray((2 * n * L,), np.uint8)
sides, pins = [], []
for text, ro, br in (side_a, side_b):
    pt = dp.PinnedArray(text.shape, np.uint8)
    pt.a[...] = text
    pins.append(pt)  # the array lives as long as its PinnedArray
    sides.append((pt.a, ro, br))
st = dp.DpStream(params, int(os.environ.get("SEGMENTS", "4")))
for what in (("columns",) if os.environ.get("ONLY_COLUMNS") else ("columns", "texts")):
    for rep in range(4):
        sys.stderr.write("-- %s, pass %d\n" % (what, rep))
        t = time.perf_counter()
        if what == "columns":
            st.align(pin, ps.a, po.a, pn.a)
        else:
            st.align_text(sides[0], sides[1], ps.a, po.a, pn.a)
        sys.stderr.write("   %.3f ms  %.0f GCUPS\n" % ((time.perf_counter() - t) * 1e3, n * L * L / (time.perf_counter() - t) / 1e9))
