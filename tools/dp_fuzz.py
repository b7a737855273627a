#!/usr/bin/env python3
"""Differential fuzz of the DP on the GPU box: random batches (pair count, lengths, depth, related or unrelated profiles, scoring)
under random kernel choices (path mode, columns per lane, waves and workgroups per pair, band, walk lanes, chunks, tiers, stripe
widths: spelt as PM_DP_* variables, which dp.options_from_env turns into the pm_dp_options_t the library is given), every score and
every path against oracle/dp_oracle.c.  python tools/dp_fuzz.py [seconds] [first seed]; prints one line per case, stops at the
first mismatch with the environment that reproduces it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np

os.environ.setdefault("PM_DP_SEGMENT_CELLS", "1")  # the host-fed engine in as many segments as drawn, however small the batch  # noqa: E402
import pyoracle  # noqa: E402
from paramugsy_amd import dp  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
KNOBS = {"PM_DP_MODE": ["ckpt", "ckpt", "bits", None], "PM_DP_COLS": ["8", "16", None], "PM_DP_WAVES": ["1", "2", "4", "8", "16", None, None],
         "PM_DP_GROUPS": ["1", "2", "4", "8", None, None], "PM_DP_BAND": ["0", "1", None], "PM_DP_WALK_LANES": ["4", "8", "16", "32", None, None],
         # the chunk pipeline (these matter for the resident batch when the case draws a small workspace budget, below): parts of the
         # workspace, the fill kernels of every part on a stream of their own or not, the gate kernel in front of them or not, a batch
         # that fits cut into chunks all the same
         "PM_DP_SLOTS": ["2", "3", "4", None, None], "PM_DP_NO_GATE": ["1", None, None],
         "PM_DP_SPLIT": ["2", "3", "5", None, None, None],
         # the tiers (a chunk's longest pairs in launches of their own, with their own path kernels) for chunks of 8 or 16 pairs already;
         # full-width last stripes instead of the narrow ones
         "PM_DP_TIER_MIN_PAIRS": ["8", "8", "16", None], "PM_DP_NO_TIERS": ["1", None, None, None, None], "PM_DP_TAIL": ["0", None, None],
         # tiles from a queue (round 5): never, tiles of 64 / 128 / 512 steps for every checkpoint or score launch, or the library's own rule
         "PM_DP_TILE": ["0", "64", "64", "128", "512", None, None],
         # the walk beside the fill kernel of its own launch (off by default): on for a quarter of the cases
         "PM_DP_EARLY_WALK": ["1", None, None, None]}


def random_case(rng):
    extreme = rng.random() < 0.2  # deep columns and large weights, near the int16 / score-bound limits of pm_dp_batch_create
    rows = int(rng.choice([100, 127, 128, 200, 255])) if extreme else int(rng.choice([1, 2, 3, 4, 8, 13, 32, 60]))
    n = int(rng.choice([1, 2, 3, 7, 20, 40]))
    longest = int(rng.choice([10, 60, 250])) if extreme else int(rng.choice([40, 300, 1100, 2500, 5000]))
    cells_left = 6e6
    la, lb = [], []
    for _ in range(n):
        a = int(rng.integers(0 if rng.random() < 0.03 else 1, longest + 1))
        b = int(rng.integers(0 if rng.random() < 0.03 else 1, longest + 1))
        if rng.random() < 0.5:  # related lengths
            b = max(1, int(a * (1 + rng.normal(0, 0.05))))
        if a * b > cells_left:
            b = max(1, int(cells_left // max(a, 1)))
        cells_left = max(cells_left - a * b, 2e4)
        la.append(a)
        lb.append(b)
    related = rng.random() < 0.6

    def picks_random(total):  # the symbol (0-3 ACGT, 4 gap) every row holds in every column
        return rng.integers(0, 5 if rng.random() < 0.5 else 4, size=(total, rows))

    def cols_of(pick):
        c = np.zeros((len(pick), 8), dtype=np.uint8)
        for s in range(5):
            c[:, s] = (pick == s).sum(axis=1)
        return c
    PA = [picks_random(x) for x in la]
    PB = []
    for k, x in enumerate(lb):
        if related and la[k] > 0 and x > 0:  # B = A resampled to lb columns with some columns replaced
            idx = np.minimum((np.arange(x) * la[k]) // x, la[k] - 1)
            b = PA[k][idx].copy()
            noise = rng.random(x) < 0.1
            b[noise] = picks_random(int(noise.sum()))
            PB.append(b)
        else:
            PB.append(picks_random(x))
    A, B = [cols_of(x) for x in PA], [cols_of(x) for x in PB]
    cat = lambda parts: np.concatenate(parts) if sum(len(p) for p in parts) else np.zeros((0, 8), np.uint8)
    inputs = dp.DpInputs(cat(A), np.concatenate([[0], np.cumsum(la)]).astype(np.int64), cat(B), np.concatenate([[0], np.cumsum(lb)]).astype(np.int64))
    letters = np.frombuffer(b"ACGT-", dtype=np.uint8)
    # the rows themselves, as MAF blocks (a profile of 0 columns is a block without rows)
    inputs.blocks = [[[letters[pk[:, r]].tobytes() for r in range(rows)] if len(pk) else [] for pk in side] for side in (PA, PB)]
    p = dp.make_params(rows, rows)
    if extreme:
        hi = int(rng.choice([1, 3, 20, 32767 // rows]))
        for k in range(25):
            p.sub[k] = int(rng.integers(-hi, hi + 1))
        p.gap_open = int(rng.integers(0, 32767))
        p.gap_extend = int(rng.integers(0, min(p.gap_open, 2000) + 1))
    elif rng.random() < 0.5:
        hi = max(1, min(6, 127 // rows)) if rng.random() < 0.5 else 6
        for k in range(25):
            p.sub[k] = int(rng.integers(-hi, hi + 1))
        p.gap_open = int(rng.integers(0, 30)) * rows
        p.gap_extend = int(rng.integers(0, 5)) * rows
    return inputs, p, rows, la, lb


t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    t_case = time.time()
    inputs, p, rows, la, lb = random_case(rng)
    env = {k: str(rng.choice([x for x in v if x is not None] + [""] * sum(x is None for x in v))) for k, v in KNOBS.items()}
    env = {k: v for k, v in env.items() if v}
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    dp.set_default_options(dp.options_from_env())  # for the batches the library makes itself (the device-list entry)
    # engines: the resident batch; the host-fed engine (upload segments, variant re-check) with packed columns or with the rows'
    # texts; the device-list entry with two to four workers on the one GPU
    draw = rng.random()
    engine = "batch" if draw >= 0.45 or inputs.n_pairs == 0 else ("stream" if draw < 0.2 else ("stream_text" if draw < 0.33 else "multi"))
    if os.environ.get("PM_FUZZ_ENGINE") and inputs.n_pairs > 0:  # one engine only (hunting)
        engine = os.environ["PM_FUZZ_ENGINE"]
    n_workers = int(rng.integers(2, 5))
    if engine == "stream_text" and rows > 255:
        engine = "stream"
    if os.environ.get("PM_FUZZ_DRY"):  # what a seed draws, without running it (no GPU needed)
        print("seed", seed, "pairs", len(la), "rows", rows, "la", la[:12], "lb", lb[:12], env, "engine", engine, "workers", n_workers, flush=True)
        seed += 1
        cases += 1
        if cases >= int(os.environ["PM_FUZZ_DRY"]):
            break
        continue
    streamed = engine != "batch"
    try:
        if engine in ("stream", "stream_text"):
            # the engine cuts a batch of at least four segments' worth of cells into a chunk per segment, ordered longest first, and
            # runs a smaller one as one chunk in the input order with a launch per segment: half of the cases each
            cells_case = int(np.sum(np.asarray(la, dtype=np.int64) * np.asarray(lb, dtype=np.int64)))
            seg_cells = 1 if rng.random() < 0.5 else max(1, cells_case // 3 + 1)
            st = dp.DpStream(p, segments=int(rng.integers(1, 7)), options=dp.options_from_env(dict(os.environ, PM_DP_SEGMENT_CELLS=str(seg_cells))))
            env = dict(env, segment_cells=seg_cells)
            try:
                if engine == "stream":
                    scores, ops, n_ops = st.align(inputs)
                else:
                    scores, ops, n_ops = st.align_text(dp.flatten_blocks(inputs.blocks[0]), dp.flatten_blocks(inputs.blocks[1]))
                    scores, n_ops = scores[:inputs.n_pairs], n_ops[:inputs.n_pairs]
            finally:
                st.close()
            paths = dp.paths_of(inputs, ops, n_ops)
            v = {"checkpoints": "?", "cols_per_lane": "?"}
        elif engine == "multi":
            scores, ops, n_ops = dp.align_multi(inputs, p, [0] * n_workers)
            paths = dp.paths_of(inputs, ops, n_ops)
            v = {"checkpoints": "?", "cols_per_lane": "?"}
        else:
            # a third of the resident batches with a workspace budget that cuts them into several chunks
            small = rng.random() < 0.33
            b = dp.DpBatch(inputs, p, tb_budget_bytes=int(rng.choice([1, 4, 16])) << 20 if small else 0)
            if small:
                env = dict(env, chunks=b.info()["chunks"])
            b.run(True)
            scores, ops, n_ops = b.fetch()
            paths = b.paths(ops, n_ops)
            v = b.variant()
            b.close()
    except Exception as exc:  # refused batches (score bound) are fine
        print("seed", seed, "refused:", str(exc)[:80], flush=True)
        seed += 1
        continue
    if streamed:
        env = dict(env, engine=engine)
    t_gpu = time.time() - t_case
    o_scores, o_paths = pyoracle.dp_align(inputs, p)
    ok = np.array_equal(scores, o_scores) and all(np.array_equal(x, y) for x, y in zip(paths, o_paths))
    if t_gpu > 0.5:  # a case that takes the GPU side this long is waiting for something (a bounded wait running out?)
        env = dict(env, SLOW_s=round(t_gpu, 2))
    print("seed", seed, "pairs", len(la), "rows", rows, "max", max(la + [0]), "x", max(lb + [0]), env, ("ckpt" if v["checkpoints"] else "bits") if v["checkpoints"] != "?" else "-", "cols", v["cols_per_lane"],
          "OK" if ok else "MISMATCH", flush=True)
    if not ok:
        bad = [k for k in range(len(la)) if scores[k] != o_scores[k] or not np.array_equal(paths[k], o_paths[k])]
        print("  first bad pairs", bad[:5], "la/lb", [(la[k], lb[k]) for k in bad[:5]], "workers", n_workers if engine == "multi" else "-", flush=True)
        for k in bad[:3]:
            diff = [i for i in range(min(len(paths[k]), len(o_paths[k]))) if paths[k][i] != o_paths[k][i]]
            print("   pair", k, "score", int(scores[k]), "oracle", int(o_scores[k]), "path length", len(paths[k]), "oracle", len(o_paths[k]),
                  "first differing op", diff[:1], "ops differing", len(diff), flush=True)
        if os.environ.get("PM_FUZZ_KEEP_GOING"):
            seed += 1
            continue
        sys.exit(1)
    cases += 1
    if os.environ.get("PM_FUZZ_SAME_SEED"):  # the same case again and again (hunting something that does not fail every time)
        continue
    seed += 1
print("cases", cases, "all equal the oracle")
