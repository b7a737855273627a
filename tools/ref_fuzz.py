#!/usr/bin/env python3
"""The three upstream binaries of the path (oracle/_ref/{m_translate, m_sort_delta, maf_analyzer}, built from /root/reference by
oracle/Makefile and carried to the GPU box with the snapshot) against the library on random inputs, byte for byte:
pm_translate_files on random workloads, pm_sort_delta on random delta text, pm_maf_analyzer on random MAF files with overlapping
rows.  python tools/ref_fuzz.py [seconds] [first seed]"""
import ctypes as C
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from paramugsy_amd import capi, synth  # noqa: E402
from paramugsy_amd.translate import translate, translate_multi  # noqa: E402
from test_translate_gpu import MODES  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
for exe in ("m_translate", "m_sort_delta", "maf_analyzer"):
    if not os.path.exists(os.path.join(REF, exe)):
        sys.exit("oracle/_ref/%s is missing: run `make oracle` where /root/reference exists" % exe)
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
tmp = tempfile.mkdtemp(prefix="reffuzz")
count = {"translate": 0, "sort": 0, "maf": 0}
out_bytes = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    kind = ("translate", "sort", "maf")[seed % 3]
    d = os.path.join(tmp, "case")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    a, b = os.path.join(d, "ref.out"), os.path.join(d, "gpu.out")
    if kind == "translate":
        mode = sorted(MODES)[int(rng.integers(0, len(MODES)))]
        kw = dict(MODES[mode])
        for key, lo, hi in (("gap_rate", 0.0, 0.15), ("indel_rate", 0.0, 0.08), ("rev_prob", 0.0, 0.6), ("delta_rev_prob", 0.0, 0.6),
                            ("edge_gap_prob", 0.0, 0.6), ("adjacent_prob", 0.0, 0.2)):
            if rng.random() < 0.5:
                kw[key] = float(rng.uniform(lo, hi))
        if mode != "long_rows" and rng.random() < 0.5:
            kw["entries_per_delta"] = int(rng.integers(5, 200))
        if rng.random() < 0.4:  # everything random: genome and block counts, block width, entry length, spacing, file count
            glen = int(rng.choice([2000, 30000, 300000]))
            kw = dict(n_left=int(rng.integers(1, 6)), n_right=int(rng.integers(1, 6)), genome_len=glen, n_blocks=int(rng.integers(1, 120)),
                      n_deltas=int(rng.integers(1, 5)), entries_per_delta=int(rng.integers(1, 300)), mean_cols=int(rng.choice([5, 40, 400, 3000])),
                      gap_rate=float(rng.choice([0.001, 0.01, 0.1])), mean_gap=float(rng.choice([1.5, 3.0, 12.0])), row_prob=float(rng.choice([0.5, 0.8, 1.0])),
                      rev_prob=float(rng.choice([0.0, 0.3, 1.0])), edge_gap_prob=float(rng.choice([0.0, 0.3, 0.9])), spacing=int(rng.choice([1, 5, 40, 400])),
                      mean_len=int(rng.choice([30, 300, 3000])), indel_rate=float(rng.choice([0.001, 0.01, 0.08])), mean_indel=float(rng.choice([1.5, 4.0, 15.0])),
                      adjacent_prob=float(rng.choice([0.0, 0.1, 0.4])), delta_rev_prob=float(rng.choice([0.0, 0.3, 1.0])), group=int(rng.integers(1, 5)))
            mode = "wild"
        if rng.random() < 0.25:  # rows of a genome that overlap and nest (the row index's binary search then meets ends out of order)
            kw["overlap_prob"] = float(rng.choice([0.05, 0.3, 0.8]))
            mode += " overlapping rows"
        w = synth.make_workload(os.path.join(d, "job"), seed, **kw)
        rc = subprocess.run([os.path.join(REF, "m_translate"), w.left_dir, w.right_dir, w.list_path, a], capture_output=True).returncode
        multi = int(rng.integers(2, 5)) if rng.random() < 0.3 else 0  # a third of the jobs over a device list (workers share the GPU)
        try:
            if multi:
                translate_multi(w.left_dir, w.right_dir, w.delta_paths, b, [0] * multi)
            else:
                translate(w.left_dir, w.right_dir, w.delta_paths, b)
            failed = False
        except capi.PmError:
            failed = True
        what = mode + (" x%d devices" % multi if multi else "")
        if rc != 0 or failed:  # input the reference dies on (the generator can write a delta file it cannot parse): the library must fail too
            print("seed", seed, kind, what, "reference exit", rc, "library", "fails" if failed else "succeeds", "AGREE" if (rc != 0) == failed else "DISAGREE",
                  flush=True)
            if (rc != 0) != failed:
                shutil.copytree(d, os.path.join(ROOT, "gpurun_out", "ref_fuzz_seed%d" % seed), dirs_exist_ok=True)
                sys.exit(1)
            # what was written before the failure: the reference dies with its stream's last buffer unflushed, so its file is a
            # prefix of the library's
            if not open(b, "rb").read().startswith(open(a, "rb").read()):
                print("seed", seed, "partial outputs differ", flush=True)
                shutil.copytree(d, os.path.join(ROOT, "gpurun_out", "ref_fuzz_seed%d" % seed), dirs_exist_ok=True)
                sys.exit(1)
            count["both fail"] = count.get("both fail", 0) + 1
            seed += 1
            continue
    elif kind == "sort":
        names_r = [["b", "a", "c", "ab"], ["r1"], ["x.1", "x.10", "x.2"]][int(rng.integers(0, 3))]
        names_q = [["y", "x"], ["q"], ["k2", "k1", "k3"]][int(rng.integers(0, 3))]
        n = int(rng.choice([0, 1, 5, 200, 3000]))
        text = synth.gen_delta_text(rng, names_r, names_q, 200000, 200000, n, mean_len=int(rng.choice([50, 600, 3000])), group=int(rng.integers(1, 5)))
        src = os.path.join(d, "in.delta")
        open(src, "w").write(text)
        with open(src, "rb") as f, open(a, "wb") as o:
            rc = subprocess.run([os.path.join(REF, "m_sort_delta")], stdin=f, stdout=o).returncode
        capi.check(capi.lib().pm_sort_delta(src.encode(), b.encode(), 0))
        what = "%d entries" % n
    else:
        genomes = [["A", "B", "C"], ["g1", "g2"], ["A", "B", "C", "D", "E"]][int(rng.integers(0, 3))]
        glen = int(rng.choice([500, 4000, 50000]))
        blocks = synth.gen_side(rng, genomes, glen, int(rng.integers(1, 60)), mean_cols=int(rng.choice([10, 80, 300])), spacing=int(rng.integers(1, 30)),
                                gap_rate=0.0, edge_gap_prob=0.0)
        if rng.random() < 0.6:  # a second set that overlaps the first
            blocks = blocks + synth.gen_side(rng, genomes[:2], glen, int(rng.integers(1, 40)), mean_cols=90, spacing=int(rng.integers(1, 40)), gap_rate=0.0,
                                             edge_gap_prob=0.0)
        order = rng.permutation(len(blocks))
        src = os.path.join(d, "in.maf")
        open(src, "w").write(synth.side_to_maf_text([blocks[i] for i in order]))
        with open(a, "wb") as o:
            rc = subprocess.run([os.path.join(REF, "maf_analyzer"), src], stdout=o).returncode
        capi.check(capi.lib().pm_maf_analyzer(src.encode(), b.encode(), 0))
        what = "%d blocks" % len(blocks)
    same = rc == 0 and open(a, "rb").read() == open(b, "rb").read()
    print("seed", seed, kind, what, "reference exit", rc, os.path.getsize(a), "bytes", "EQUAL" if same else "DIFFERENT", flush=True)
    if not same:
        keep = os.path.join(ROOT, "gpurun_out", "ref_fuzz_seed%d" % seed)
        shutil.copytree(d, keep, dirs_exist_ok=True)
        sys.exit(1)
    count[kind] += 1
    out_bytes += os.path.getsize(a)
    seed += 1
shutil.rmtree(tmp, ignore_errors=True)
print("cases", count, "reference output bytes compared", out_bytes, ": all equal")
