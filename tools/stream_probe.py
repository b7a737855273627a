#!/usr/bin/env python3
"""The host-fed engine on one shape: python3 tools/stream_probe.py SPEC [SEGMENTS ...]   (SPEC as tools/shares_probe.py's)
Per number of upload segments: ms and GCUPS of pm_dp_stream_align from pinned host columns to pinned results (best of 3 after one
warm-up), beside the resident batch's step.  PM_TIMING=1 prints the engine's phases."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from paramugsy_amd import dp  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from shares_probe import make  # noqa: E402


def main():
    trace = sys.argv[1] == "--trace"  # two calls of the engine and nothing else (tools/trace_stream.sh)
    if trace:
        del sys.argv[1]
    spec = sys.argv[1]
    segs = [int(x) for x in sys.argv[2:]] or [4]
    inputs, rows = make(spec)
    params = dp.make_params(rows, rows)
    n = inputs.n_pairs
    if trace:
        pa, pb = dp.PinnedArray(inputs.cols_a.shape, np.uint8), dp.PinnedArray(inputs.cols_b.shape, np.uint8)
        pa.a[...] = inputs.cols_a
        pb.a[...] = inputs.cols_b
        pin = dp.DpInputs(pa.a, inputs.off_a, pb.a, inputs.off_b)
        ps, pn = dp.PinnedArray((n,), np.int32), dp.PinnedArray((n,), np.int32)
        po = dp.PinnedArray((max(1, int(inputs.off_a[-1] + inputs.off_b[-1])),), np.uint8)
        st = dp.DpStream(params, segs[0])
        for _ in range(2):
            st.align(pin, ps.a, po.a, pn.a)
            time.sleep(0.05)
        st.close()
        return
    b = dp.DpBatch(inputs, params)
    for _ in range(2):
        b.run(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        b.run(True)
    torch.cuda.synchronize()
    step = (time.perf_counter() - t0) / 3
    cells = float(np.sum((inputs.off_a[1:] - inputs.off_a[:-1]).astype(np.float64) * (inputs.off_b[1:] - inputs.off_b[:-1])))
    print("%-14s resident step %.1f ms  %.0f GCUPS" % (spec, step * 1e3, cells / step / 1e9), flush=True)
    b.close()
    pa, pb = dp.PinnedArray(inputs.cols_a.shape, np.uint8), dp.PinnedArray(inputs.cols_b.shape, np.uint8)
    pa.a[...] = inputs.cols_a
    pb.a[...] = inputs.cols_b
    pin = dp.DpInputs(pa.a, inputs.off_a, pb.a, inputs.off_b)
    ps, pn = dp.PinnedArray((n,), np.int32), dp.PinnedArray((n,), np.int32)
    po = dp.PinnedArray((max(1, int(inputs.off_a[-1] + inputs.off_b[-1])),), np.uint8)
    print("  columns up %.0f MB, results down %.0f MB" % ((pa.a.nbytes + pb.a.nbytes) / 1e6, (po.a.nbytes + 8 * n) / 1e6))
    for s in segs:
        st = dp.DpStream(params, s)
        st.align(pin, ps.a, po.a, pn.a)
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            st.align(pin, ps.a, po.a, pn.a)
            best = min(best, time.perf_counter() - t0)
        print("  segments %3d: %.1f ms  %.0f GCUPS" % (s, best * 1e3, cells / best / 1e9), flush=True)
        st.close()
    for x in (pa, pb, ps, pn, po):
        x.close()


if __name__ == "__main__":
    main()
