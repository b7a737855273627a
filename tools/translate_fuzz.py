#!/usr/bin/env python3
"""Differential fuzz of the translate path on the GPU box: random workloads (the parameter sets of tests/test_translate_gpu.py with
random perturbations, half of them with the tables made inconsistent so that units fail) through pm_job_* against
oracle/pm_oracle.cc (which tests/test_oracle_vs_ref.py and the goldens pin to the upstream binary): status per unit, entries and
offsets must be equal; and the unit list the device makes for the loaded files (pm_job_create_from_workload) against the host's.  python tools/translate_fuzz.py [seconds] [first seed]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import pyoracle  # noqa: E402
from paramugsy_amd import synth  # noqa: E402
from paramugsy_amd.translate import TranslateJob, Workload  # noqa: E402
from test_translate_gpu import MODES, corrupt_tables  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
cases = units = failed_units = 0
tmp = tempfile.mkdtemp(prefix="trfuzz")
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    mode = sorted(MODES)[int(rng.integers(0, len(MODES)))]
    kw = dict(MODES[mode])
    for key, lo, hi in (("gap_rate", 0.0, 0.15), ("indel_rate", 0.0, 0.08), ("rev_prob", 0.0, 0.6), ("delta_rev_prob", 0.0, 0.6),
                        ("edge_gap_prob", 0.0, 0.6), ("adjacent_prob", 0.0, 0.2)):
        if rng.random() < 0.5:
            kw[key] = float(rng.uniform(lo, hi))
    if mode != "long_rows" and rng.random() < 0.5:
        kw["entries_per_delta"] = int(rng.integers(5, 200))
    d = os.path.join(tmp, "job")
    shutil.rmtree(d, ignore_errors=True)
    if seed >= 1000 and np.random.default_rng(seed ^ 0x5EED).random() < 0.25:  # (the seeds below 1000 are pinned by tests as they were)
        kw["overlap_prob"] = float(np.random.default_rng(seed ^ 0x5EED).choice([0.05, 0.3, 0.8]))
        mode += " overlapping rows"
    w = synth.make_workload(d, seed, **kw)
    wl = Workload.load(w.left_dir, w.right_dir, w.delta_paths)
    t = wl.tables()
    # the unit list made on the device (what the file-level entries run on) against the host's loops
    listed = TranslateJob.from_workload(wl)
    dev_units = listed.units()
    listed.close()
    if any(not np.array_equal(dev_units[k], t.units[k]) for k in ("delta", "left", "right")):
        print("seed", seed, mode, "the device's unit list differs from the host's: UNIT LIST MISMATCH", flush=True)
        sys.exit(1)
    corrupt = rng.random() < 0.5
    if corrupt:
        corrupt_tables(t, rng)
    # round 5: four cases in ten lie far along their sequences (positions that need 64 bits; the library keeps them, and only them, wide)
    far = np.random.default_rng(seed ^ 0xFA2)
    where = ""
    if seed >= 1000 and far.random() < 0.4:
        shifts = [int(far.choice([(1 << 25) - 100000, (1 << 25) + 3, 1 << 31, (1 << 40) + 17, (1 << 61) + 5, 0])) for _ in range(2)]
        t = synth.shift_positions(t, shifts[0], shifts[1])
        where = " moved by %d / %d" % tuple(shifts)
    job = TranslateJob(t)
    where += " bits %d/%d" % (job.coordinate_bits(), job.position_bits())
    job.run()
    res = job.fetch()
    job.close()
    ora = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
    # per unit: same status, same entries, same offsets
    eo_g, eo_o = res.unit_entry_off, ora["unit_entry_off"]
    ok = True
    for u in range(t.n_units):
        sg, so = int(res.status[u]), int(ora["status"][u])
        ng, no = int(eo_g[u + 1] - eo_g[u]), int(eo_o[u + 1] - eo_o[u])
        same = sg == so and ng == no
        for e in range(ng if same else 0):
            eg, eo_ = res.entries[int(eo_g[u]) + e], ora["entries"][int(eo_o[u]) + e]
            same = same and all(int(eg[k]) == int(eo_[k]) for k in ("ref_start", "ref_end", "qry_start", "qry_end", "n_offsets"))
            if same:
                a_, b_ = int(eg["offset_begin"]), int(eo_["offset_begin"])
                same = np.array_equal(res.offsets[a_:a_ + int(eg["n_offsets"])], ora["offsets"][b_:b_ + int(eo_["n_offsets"])])
        if not same:
            ok = False
            print("  unit", u, "status gpu", sg, "oracle", so, "entries gpu", ng, "oracle", no, flush=True)
            break
    print("seed", seed, mode + where, "corrupt" if corrupt else "clean", "units", t.n_units, "failing", int((res.status != 0).sum()), "entries", len(res.entries), "OK" if ok else "MISMATCH", flush=True)
    if not ok:
        sys.exit(1)
    cases += 1
    units += t.n_units
    failed_units += int((res.status != 0).sum())
    seed += 1
shutil.rmtree(tmp, ignore_errors=True)
print("cases", cases, "units", units, "of which failing (with the reference's failure class)", failed_units, ": every unit equals the oracle")
