#!/usr/bin/env python3
"""A few passes of one DP shape (for rocprofv3): python3 tools/dp_run_once.py pairs:rows:len [reps]; PM_DP_* env selects variants."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from paramugsy_amd import dp  # noqa: E402
n, rows, L = (int(x) for x in sys.argv[1].split(":"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
inputs = dp.synth_pairs_fast(20261003, n, rows, L) if n * L <= 20000000 else dp.synth_batch(20261003, [L] * n, [L] * n, rows, rows)
b = dp.DpBatch(inputs, dp.make_params(rows, rows))
paths = os.environ.get("RUN_ONCE_SCORES_ONLY") != "1"  # RUN_ONCE_SCORES_ONLY=1: passes without the path (no checkpoints written)
for _ in range(reps):
    b.run(paths)
b.fetch()
b.close()
