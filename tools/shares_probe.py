#!/usr/bin/env python3
"""What one GPU does with the shares a strong split hands it: python3 tools/shares_probe.py SPEC [SPEC ...]

SPEC = ns:N | c1:N | c2:N | deep:N | u:N:ROWS:LEN   (N pairs of bench.py's configuration, or a uniform batch)
Per spec: GCUPS and ms of a whole step (fill + path kernels, 5 timed passes after 2 warm-ups), the fill and path kernels' own
device times (HIP events), the time of a scores-only pass, chunks and the kernel variant.  PM_DP_* select variants."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from paramugsy_amd import dp  # noqa: E402
from paramugsy_amd.synth_device import synth_batch_device  # noqa: E402

SHAPES = {"ns": (8, 4096), "c1": (2, 1000), "c2": (4, 0), "deep": (32, 10000)}


def make(spec):
    f = spec.split(":")
    n = int(f[1])
    rows, L = (int(f[2]), int(f[3])) if f[0] == "u" else SHAPES[f[0]]
    if L > 0:
        la = lb = np.full(n, L, dtype=np.int64)
    else:
        la, lb = dp.ragged_lengths(20261003, n)
    return synth_batch_device(20261003 * 1000003, la, lb, rows, rows), rows


def main():
    if sys.argv[1] == "--trace":  # two passes of one shape and nothing else (tools/trace_once.sh)
        inputs, rows = make(sys.argv[2])
        b = dp.DpBatch(inputs, dp.make_params(rows, rows))
        if os.environ.get("PROBE_SYNC", "1") == "1":
            for _ in range(2):
                b.run(True)
                torch.cuda.synchronize()
                time.sleep(0.01)
        else:  # passes back to back, as bench.py's timed region issues them
            b.run(True)
            torch.cuda.synchronize()
            time.sleep(0.01)
            for _ in range(3):
                b.run(True)
            torch.cuda.synchronize()
        b.fetch(with_paths=False)
        b.close()
        return
    reps = int(os.environ.get("PROBE_REPS", "5"))
    print("%-16s %8s %9s %9s %9s %10s %6s  %s" % ("spec", "GCUPS", "ms/step", "fill ms", "path ms", "scores ms", "chunks", "variant"))
    own_stream = torch.cuda.Stream() if os.environ.get("PROBE_STREAM") else None  # else the null stream
    stream = own_stream.cuda_stream if own_stream else 0
    for spec in sys.argv[1:]:
        inputs, rows = make(spec)
        b = dp.DpBatch(inputs, dp.make_params(rows, rows))
        for _ in range(2):
            b.run(True, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            b.run(True, stream)
            if os.environ.get("PROBE_SYNC"):  # one pass at a time
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        prof = [b.run_profiled(True) for _ in range(3)]
        b.run(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            b.run(False)
        torch.cuda.synchronize()
        ms_scores = (time.perf_counter() - t0) / reps * 1e3
        v = b.variant()
        print("%-16s %8.0f %9.3f %9.3f %9.3f %10.3f %6d  C%d dot4=%d uni=%d ckpt=%d" % (
            spec, inputs.cells / ms / 1e6, ms, min(p[0] for p in prof), min(p[1] for p in prof), ms_scores, b.info()["chunks"],
            v["cols_per_lane"], v["dot4"], v["uniform_depth"], v["checkpoints"]), flush=True)
        b.fetch(with_paths=False)
        b.close()
        del b, inputs


if __name__ == "__main__":
    main()
