#!/usr/bin/env python3
"""GPU box: where do tiled and untiled fills differ?  Per tile size and pair: score / path against the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch  # noqa
from paramugsy_amd import dp
import pyoracle
la = np.array([1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 500, 700, 1000, 0, 300, 257, 640, 705])
lb = np.array([2100, 1024, 1025, 3000, 5000, 100, 2049, 4097, 1023, 1500, 2600, 4200, 1300, 50, 0, 3073, 2048, 1100])
inputs = dp.synth_batch(77, la, lb, 3, 4)
params = dp.make_params(3, 4)
o_scores, o_paths = pyoracle.dp_align(inputs, params)
for dot4 in (1, 0):
    for tile in (1, 64, 128, 320, 1024):
        for rep in range(2):
            b = dp.DpBatch(inputs, params, options=dp.options(path_mode=2, tile_steps=tile, int16_weights=1 - dot4, full_stripes=1, cols_per_lane=16))
            b.run(True)
            s, ops, n = b.fetch()
            paths = b.paths(ops, n)
            bad_s = [k for k in range(len(la)) if s[k] != o_scores[k]]
            bad_p = [k for k in range(len(la)) if not np.array_equal(paths[k], o_paths[k])]
            detail = ""
            for k in bad_p[:2]:
                m = min(len(paths[k]), len(o_paths[k]))
                d = np.nonzero(paths[k][:m] != o_paths[k][:m])[0]
                detail += " pair %d (%dx%d): len %d/%d first diff at op %s" % (k, la[k], lb[k], len(paths[k]), len(o_paths[k]), d[:1])
            print("dot4=%d tile=%4d rep %d: bad scores %s bad paths %s%s" % (dot4, tile, rep, bad_s, bad_p, detail), flush=True)
            b.close()
# one pair alone
for k in (12, 11):
    one = dp.synth_batch(77, la[k:k + 1], lb[k:k + 1], 3, 4)
    os1, op1 = pyoracle.dp_align(one, params)
    for tile in (1, 64, 128):
        b = dp.DpBatch(one, params, options=dp.options(path_mode=2, tile_steps=tile, int16_weights=1, full_stripes=1, cols_per_lane=16, band=1))
        b.run(True)
        s, ops, n = b.fetch()
        p = b.paths(ops, n)
        print("alone pair %d tile %d: score ok %s path ok %s" % (k, tile, s[0] == os1[0], np.array_equal(p[0], op1[0])), flush=True)
        b.close()
