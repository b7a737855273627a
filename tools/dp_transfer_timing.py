import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from paramugsy_amd import dp
inputs = dp.synth_pairs_fast(1, 10000, 2, 1000)
params = dp.make_params(2, 2)
b = dp.DpBatch(inputs, params); b.run(traceback=True); b.fetch(); b.close()   # warm
for rep in range(3):
    t0 = time.perf_counter()
    b = dp.DpBatch(inputs, params)
    t1 = time.perf_counter()
    b.run(traceback=True)
    s, ops, n = b.fetch()
    t2 = time.perf_counter()
    b.close()
    t3 = time.perf_counter()
    print("create(upload+alloc) %.1f ms  run+fetch %.1f ms  close %.1f ms  -> %.0f GCUPS incl. transfers" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, 1e10/(t2-t0)/1e9))
