#!/usr/bin/env python3
"""`mugsy_profiles make` and `untranslate` on the GPU against their Python transcriptions (oracle/make_oracle.py,
oracle/untranslate_oracle.py: "restated from source, not executed" -- the OCaml cannot run here) on random inputs, byte for byte:
random MAF sides (genome count, block count and width, gap and edge-gap rates, both strands) through pm_profiles_make, then a fake
mugsy MAF over random column ranges of the resulting profiles through pm_untranslate.  python tools/stage_fuzz.py [seconds] [seed]"""
import ctypes as C
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import make_oracle  # noqa: E402
import untranslate_oracle as uo  # noqa: E402
from paramugsy_amd import capi, synth  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
tmp = tempfile.mkdtemp(prefix="stagefuzz")
cases = rows_out = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    d = os.path.join(tmp, "case")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    sets = []
    ok = True
    for side in ("l", "r"):
        genomes = ["%s%d.c" % (side.upper(), k) for k in range(int(rng.integers(1, 6)))]
        blocks = synth.gen_side(rng, genomes, int(rng.choice([800, 20000, 200000])), int(rng.integers(1, 40)), mean_cols=int(rng.choice([6, 150, 900])),
                                gap_rate=float(rng.choice([0.0, 0.04, 0.2])), edge_gap_prob=float(rng.choice([0.0, 0.4, 0.9])),
                                rev_prob=float(rng.choice([0.0, 0.5])))
        maf = synth.side_to_maf_text(blocks)
        src = os.path.join(d, side + ".maf")
        open(src, "w").write(maf)
        out = os.path.join(d, side)
        os.makedirs(out)
        capi.check(capi.lib().pm_profiles_make(src.encode(), out.encode(), side.encode(), 0))
        prof, fasta = make_oracle.make(maf, side)
        ok = ok and open(os.path.join(out, "profiles")).read() == prof and open(os.path.join(out, "sequences.fasta")).read() == fasta
        sets.append((side, blocks, prof, out))
    lines = ["##maf version=1 scoring=mugsy", "# produced by a fake mugsyWGA"]
    for _ in range(int(rng.integers(1, 60))):
        lines.append("a score=%d label=1 mult=2" % int(rng.integers(0, 999)))
        for _r in range(int(rng.integers(1, 4))):
            side, blocks, _p, _o = sets[int(rng.integers(0, 2))]
            if not blocks:
                continue
            b = int(rng.integers(0, len(blocks)))
            cols = len(blocks[b].rows[0].text)
            size = int(rng.integers(1, cols + 1))
            start = int(rng.integers(0, cols - size + 1))
            text = list("ACGT"[int(x)] for x in rng.integers(0, 4, size=size))
            for _g in range(int(rng.integers(0, 4))):
                text.insert(int(rng.integers(0, len(text) + 1)), "-")
            strand = "+" if rng.random() < 0.6 else "-"
            lines.append("s %s.%s_%04d %d %d %s %d %s" % (side, side, b, start, size, strand, cols, "".join(text)))
        lines.append("")
    maf = "\n".join(lines) + "\n"
    open(os.path.join(d, "in.maf"), "w").write(maf)
    dirs = [s[3].encode() for s in sets]
    arr = (C.c_char_p * len(dirs))(*dirs)
    capi.check(capi.lib().pm_untranslate(arr, len(dirs), os.path.join(d, "in.maf").encode(), os.path.join(d, "out.maf").encode(), 0))
    want = uo.untranslate([s[2] for s in sets], maf)
    got = open(os.path.join(d, "out.maf")).read()
    ok2 = got == want
    print("seed", seed, "blocks", [len(s[1]) for s in sets], "make", "EQUAL" if ok else "DIFFERENT", "untranslate", want.count("\ns "), "rows",
          "EQUAL" if ok2 else "DIFFERENT", flush=True)
    if not (ok and ok2):
        shutil.copytree(d, os.path.join(ROOT, "gpurun_out", "stage_fuzz_seed%d" % seed), dirs_exist_ok=True)
        sys.exit(1)
    cases += 1
    rows_out += want.count("\ns ")
    seed += 1
shutil.rmtree(tmp, ignore_errors=True)
print("cases", cases, "(two make runs and one untranslate each), untranslated rows", rows_out, ": all equal the transcriptions")
