#!/usr/bin/env python3
"""bits vs checkpoints vs what dp_batch_plan picks, over small and mid-size batches (GPU box):
python tools/dp_mode_sweep.py [pairs:rows:len ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paramugsy_amd import dp  # noqa: E402

DEFAULT = ["1:8:4096", "8:8:4096", "32:8:4096", "64:8:4096", "128:8:4096", "256:8:4096", "512:8:4096", "1024:8:4096", "16:2:1000",
           "64:2:1000", "256:2:1000", "1000:2:1000", "2000:2:1000", "3000:2:1000", "1:32:10000", "8:32:10000", "32:32:10000",
           "128:32:10000", "64:4:300", "1024:4:300", "8192:4:300"]


def total(inputs, params, env):
    for k, v in env.items():
        os.environ[k] = v
    b = dp.DpBatch(inputs, params)
    for k in env:
        del os.environ[k]
    b.run_profiled(True)
    t = min(sum(b.run_profiled(True)) for _ in range(5))
    v = b.variant()
    b.close()
    return t, v


for sh in sys.argv[1:] or DEFAULT:
    n, rows, L = (int(x) for x in sh.split(":"))
    inputs = dp.synth_pairs_fast(20261003, n, rows, L) if n * L <= 20000000 else dp.synth_batch(20261003, [L] * n, [L] * n, rows, rows)
    params = dp.make_params(rows, rows)
    tb, _ = total(inputs, params, {"PM_DP_MODE": "bits"})
    tc, _ = total(inputs, params, {"PM_DP_MODE": "ckpt"})
    tn, _ = total(inputs, params, {"PM_DP_MODE": "ckpt", "PM_DP_BAND": "0"})
    ta, va = total(inputs, params, {})
    best = min(tb, tc)
    print(f"{sh:>14}  bits {tb:8.3f}  ckpt {tc:8.3f}  ckpt-noband {tn:8.3f}  auto {ta:8.3f} ({'ckpt' if va['checkpoints'] else 'bits'}, cols {va['cols_per_lane']})"
          f"  {'OK' if ta <= 1.1 * best else 'MISS'}", flush=True)
