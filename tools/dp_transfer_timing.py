#!/usr/bin/env python3
"""Transfer-inclusive DP rate: host buffers in, host buffers out.  Serial (pm_dp_batch_create + run + fetch) against the
streamed engine (pm_dp_stream_align, pinned buffers) at several slice counts.
python tools/dp_transfer_timing.py [pairs:rows:len ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from paramugsy_amd import dp  # noqa: E402


def main():
    for sh in sys.argv[1:] or ["10000:2:1000"]:
        n, rows, L = (int(x) for x in sh.split(":"))
        src = dp.synth_pairs_fast(20261003, n, rows, L) if n * L <= 20000000 else dp.synth_batch(20261003, [L] * n, [L] * n, rows, rows)
        params = dp.make_params(rows, rows)
        cells = src.cells
        # serial, pageable buffers
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            b = dp.DpBatch(src, params)
            b.run(True)
            scores, ops, n_ops = b.fetch()
            best = min(best, time.perf_counter() - t0)
            b.close()
        print(json.dumps({"shape": sh, "mode": "serial create+run+fetch (pageable)", "ms": round(best * 1e3, 2), "gcups": round(cells / best / 1e9, 1)}), flush=True)
        pa, pb = dp.PinnedArray(src.cols_a.shape, np.uint8), dp.PinnedArray(src.cols_b.shape, np.uint8)
        pa.a[...] = src.cols_a
        pb.a[...] = src.cols_b
        inputs = dp.DpInputs(pa.a, src.off_a, pb.a, src.off_b)
        ps, po, pn = dp.PinnedArray((n,), np.int32), dp.PinnedArray((2 * n * L,), np.uint8), dp.PinnedArray((n,), np.int32)
        for slices in (1, 2, 3, 4, 6, 8, 12):
            st = dp.DpStream(params, slices)
            st.align(inputs, ps.a, po.a, pn.a)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                st.align(inputs, ps.a, po.a, pn.a)
                best = min(best, time.perf_counter() - t0)
            ok = bool(np.array_equal(ps.a, scores) and np.array_equal(pn.a, n_ops))
            print(json.dumps({"shape": sh, "mode": "stream, pinned, %d segments" % slices, "ms": round(best * 1e3, 2),
                              "gcups": round(cells / best / 1e9, 1), "equal": ok}), flush=True)
            st.close()
        for x in (pa, pb, ps, po, pn):
            x.close()


if __name__ == "__main__":
    main()
