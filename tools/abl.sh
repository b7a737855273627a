for sh in 64:2:1000 256:2:1000 1000:2:1000 64:8:4096 256:8:4096 32:32:10000 128:32:10000; do for c in 16 8; do
PM_DP_COLS=$c python3 - $sh <<'PY'
import sys, os
sys.path.insert(0, '.')
from paramugsy_amd import dp
n, rows, L = (int(x) for x in sys.argv[1].split(':'))
inputs = dp.synth_pairs_fast(20261003, n, rows, L) if n * L <= 20000000 else dp.synth_batch(20261003, [L]*n, [L]*n, rows, rows)
b = dp.DpBatch(inputs, dp.make_params(rows, rows))
b.run_profiled(True)
r = [b.run_profiled(True) for _ in range(5)]
t = min(x[0] + x[1] for x in r)
print(sys.argv[1], 'cols', os.environ['PM_DP_COLS'], 'ckpt' if b.variant()['checkpoints'] else 'bits', round(inputs.cells / t / 1e6, 1), 'GCUPS', round(t, 3), 'ms')
PY
done; done
