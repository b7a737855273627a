#!/usr/bin/env python3
"""Device time of the DP's two path modes (PM_DP_MODE=bits|ckpt) and of the checkpoint walk's group sizes on the bench
shapes.  Runs on the GPU box: python tools/dp_mode_timing.py [--check] [--lanes 0,8,16] [shape ...], shape = pairs:rows:len.
PM_LIB_PATH selects a build with another block geometry (-DDP_CK_R / -DDP_CK_W)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
from paramugsy_amd import dp  # noqa: E402


def one(inputs, params, env, reps=5):
    for k, v in env.items():
        os.environ[k] = v
    b = dp.DpBatch(inputs, params)
    for k in env:
        del os.environ[k]
    b.run_profiled(True)
    r = [b.run_profiled(True) for _ in range(reps)]
    so = [b.run_profiled(False) for _ in range(reps)]
    fill = min(x[0] for x in r)
    tr = min(x[1] for x in r)
    info = b.info()
    b.close()
    return {"env": env, "fill_ms": round(fill, 3), "path_ms": round(tr, 3), "score_only_ms": round(min(x[0] for x in so), 3),
            "gcups": round(info["cells"] / (fill + tr) / 1e6, 1), "chunks": info["chunks"], "GB": round(info["traceback_bytes"] / 1e9, 2)}


def check(lanes):
    import pyoracle
    inputs = dp.synth_pairs(4242, 40, 3, 700, indel_rate=0.03, vary_length=True)
    params = dp.make_params(3, 3)
    o_scores, o_paths = pyoracle.dp_align(inputs, params)
    for ln in lanes:
        os.environ["PM_DP_MODE"] = "ckpt"
        os.environ["PM_DP_WALK_LANES"] = ln
        b = dp.DpBatch(inputs, params)
        b.run(True)
        scores, ops, n_ops = b.fetch()
        ok = bool(np.array_equal(scores, o_scores)) and all(np.array_equal(p, q) for p, q in zip(b.paths(ops, n_ops), o_paths))
        print("check lanes", ln, "OK" if ok else "MISMATCH", flush=True)
        b.close()
    del os.environ["PM_DP_MODE"], os.environ["PM_DP_WALK_LANES"]


def main():
    args = sys.argv[1:]
    lanes = ["0"]
    do_check = False
    shapes = []
    while args:
        a = args.pop(0)
        if a == "--check":
            do_check = True
        elif a == "--lanes":
            lanes = args.pop(0).split(",")
        else:
            shapes.append(a)
    print("lib", os.environ.get("PM_LIB_PATH", "default"), flush=True)
    if do_check:
        check(lanes)
    for sh in shapes or ["10000:2:1000", "1536:8:4096"]:
        n, rows, L = (int(x) for x in sh.split(":"))
        inputs = dp.synth_pairs_fast(20261003, n, rows, L)
        params = dp.make_params(rows, rows)
        print("shape", sh, flush=True)
        print(json.dumps(one(inputs, params, {"PM_DP_MODE": "bits"})), flush=True)
        for ln in lanes:
            print(json.dumps(one(inputs, params, {"PM_DP_MODE": "ckpt", "PM_DP_WALK_LANES": ln})), flush=True)


if __name__ == "__main__":
    main()
