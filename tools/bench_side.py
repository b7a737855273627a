#!/usr/bin/env python3
"""Measurement of the four stage tools either side of the translate path (SURVEY 8 "next" rows): m_sort_delta,
maf_analyzer, `mugsy_profiles make`, `mugsy_profiles untranslate`.  Run on the GPU box:

    python tools/bench_side.py gen  <dir>      # synthetic inputs (sizes below), written once
    python tools/bench_side.py run  <dir>      # wall time of each GPU executable and of its CPU counterpart, byte comparison
    (kernel times: `rocprofv3 --kernel-trace --stats -- bin/<tool> ...` on the same inputs; tools/refresh_side.sh)

CPU counterparts: the upstream binaries oracle/_ref/{m_sort_delta,maf_analyzer} (kind "reference"); for make and
untranslate the reference is OCaml, which this image cannot build, so the counterpart is this repo's Python restatement
(oracle/make_oracle.py, oracle/untranslate_oracle.py; kind "port", on a bounded sample: pure Python is slow).
One JSON document on stdout."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from paramugsy_amd import synth  # noqa: E402

SORT_ENTRIES = 300000
MAF_BLOCKS = 20000
MAKE_BLOCKS = 4000
UNTR_BLOCKS = 3000
SAMPLE_BLOCKS = 150  # what the pure-Python counterparts of make/untranslate are timed on


def untranslate_maf(rng, side, blocks, n_blocks):
    """A MAF as mugsyWGA would write it over profile blocks: rows name `<side>.<side>_%04d`, coordinates are columns."""
    lines = ["##maf version=1 scoring=mugsy"]
    for _ in range(n_blocks):
        lines.append("a score=%d label=1 mult=2" % int(rng.integers(0, 999)))
        for _r in range(2):
            b = int(rng.integers(0, len(blocks)))
            cols = len(blocks[b].rows[0].text)
            size = int(rng.integers(1, cols + 1))
            start = int(rng.integers(0, cols - size + 1))
            text = "".join("ACGT"[int(x)] for x in rng.integers(0, 4, size=size))
            strand = "+" if rng.random() < 0.6 else "-"
            lines.append("s %s.%s_%04d %d %d %s %d %s" % (side, side, b, start, size, strand, cols, text))
        lines.append("")
    return "\n".join(lines) + "\n"


def gen(d):
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(20261003)
    names = ["g%d.chr" % k for k in range(8)]
    with open(os.path.join(d, "sort_in.delta"), "w") as f:
        f.write(synth.gen_delta_text(rng, names, names, 2000000, 2000000, SORT_ENTRIES, mean_len=1500, group=3))
    blocks = synth.gen_side(rng, names[:4], 8000000, MAF_BLOCKS, mean_cols=300, spacing=30)
    with open(os.path.join(d, "analyze_in.maf"), "w") as f:
        f.write(synth.side_to_maf_text(blocks))
    mk = synth.gen_side(rng, names[:4], 6000000, MAKE_BLOCKS, mean_cols=1200, spacing=60)
    with open(os.path.join(d, "make_in.maf"), "w") as f:
        f.write(synth.side_to_maf_text(mk))
    with open(os.path.join(d, "make_sample.maf"), "w") as f:
        f.write(synth.side_to_maf_text(mk[:SAMPLE_BLOCKS]))
    # untranslate reads the `profiles` file a make run wrote; write it directly in the same format
    pdir = os.path.join(d, "untr_profiles")
    synth.write_side(pdir, mk[:UNTR_BLOCKS], "l")
    with open(os.path.join(d, "untr_dirs.list"), "w") as f:
        f.write(pdir + "\n")
    with open(os.path.join(d, "untr_in.maf"), "w") as f:
        f.write(untranslate_maf(rng, "l", mk[:UNTR_BLOCKS], 40000))
    sdir = os.path.join(d, "untr_profiles_sample")
    synth.write_side(sdir, mk[:SAMPLE_BLOCKS], "l")
    with open(os.path.join(d, "untr_sample.maf"), "w") as f:
        f.write(untranslate_maf(rng, "l", mk[:SAMPLE_BLOCKS], 1500))
    print(json.dumps({k: os.path.getsize(os.path.join(d, k)) for k in sorted(os.listdir(d)) if os.path.isfile(os.path.join(d, k))}))


def timed(cmd, stdin=None, stdout=None):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, stdin=stdin, stdout=stdout, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError("%s failed: %s" % (cmd, r.stderr[-400:]))
    return dt


def same(a, b):
    return open(a, "rb").read() == open(b, "rb").read()


def run(d):
    out = {}
    binp = lambda n: os.path.join(ROOT, "bin", n)  # noqa: E731
    refp = lambda n: os.path.join(ROOT, "oracle", "_ref", n)  # noqa: E731
    # warm the GPU runtime / page cache once so that the first tool is not charged for it
    timed([binp("maf_analyzer"), os.path.join(ROOT, "tests", "golden", "highly_stitchable.maf")], stdout=subprocess.DEVNULL)

    src = os.path.join(d, "sort_in.delta")
    with open(src, "rb") as i, open(os.path.join(d, "sort_gpu.out"), "wb") as o:
        g = timed([binp("m_sort_delta")], stdin=i, stdout=o)
    with open(src, "rb") as i, open(os.path.join(d, "sort_cpu.out"), "wb") as o:
        c = timed([refp("m_sort_delta")], stdin=i, stdout=o)
    out["m_sort_delta"] = {"entries": SORT_ENTRIES, "input_bytes": os.path.getsize(src), "gpu_cli_s": g, "cpu_s": c,
                           "cpu_kind": "reference", "identical": same(os.path.join(d, "sort_gpu.out"), os.path.join(d, "sort_cpu.out"))}

    src = os.path.join(d, "analyze_in.maf")
    with open(os.path.join(d, "maf_gpu.out"), "wb") as o:
        g = timed([binp("maf_analyzer"), src], stdout=o)
    with open(os.path.join(d, "maf_cpu.out"), "wb") as o:
        c = timed([refp("maf_analyzer"), src], stdout=o)
    out["maf_analyzer"] = {"blocks": MAF_BLOCKS, "input_bytes": os.path.getsize(src), "gpu_cli_s": g, "cpu_s": c, "cpu_kind": "reference",
                           "identical": same(os.path.join(d, "maf_gpu.out"), os.path.join(d, "maf_cpu.out"))}

    import make_oracle
    src = os.path.join(d, "make_in.maf")
    od = os.path.join(d, "make_out")
    os.makedirs(od, exist_ok=True)
    g = timed([binp("mugsy_profiles"), "make", "-in_maf", src, "-out_dir", od, "-basename", "l"])
    sd = os.path.join(d, "make_sample_out")
    os.makedirs(sd, exist_ok=True)
    timed([binp("mugsy_profiles"), "make", "-in_maf", os.path.join(d, "make_sample.maf"), "-out_dir", sd, "-basename", "l"])
    t0 = time.perf_counter()
    prof, fasta = make_oracle.make(open(os.path.join(d, "make_sample.maf")).read(), "l")
    c = time.perf_counter() - t0
    ok = prof == open(os.path.join(sd, "profiles")).read() and fasta == open(os.path.join(sd, "sequences.fasta")).read()
    out["mugsy_profiles make"] = {"blocks": MAKE_BLOCKS, "input_bytes": os.path.getsize(src), "gpu_cli_s": g,
                                  "cpu_s_on_sample": c, "sample_blocks": SAMPLE_BLOCKS, "cpu_kind": "port (Python restatement)",
                                  "identical_on_sample": bool(ok)}

    import untranslate_oracle as uo
    src = os.path.join(d, "untr_in.maf")
    g = timed([binp("mugsy_profiles"), "untranslate", "-profile_paths_list", os.path.join(d, "untr_dirs.list"), "-in_maf", src, "-out_maf",
               os.path.join(d, "untr_gpu.maf")])
    with open(os.path.join(d, "untr_sample.list"), "w") as f:
        f.write(os.path.join(d, "untr_profiles_sample") + "\n")
    timed([binp("mugsy_profiles"), "untranslate", "-profile_paths_list", os.path.join(d, "untr_sample.list"), "-in_maf",
           os.path.join(d, "untr_sample.maf"), "-out_maf", os.path.join(d, "untr_sample_gpu.maf")])
    t0 = time.perf_counter()
    exp = uo.untranslate([open(os.path.join(d, "untr_profiles_sample", "profiles")).read()], open(os.path.join(d, "untr_sample.maf")).read())
    c = time.perf_counter() - t0
    out["mugsy_profiles untranslate"] = {"maf_blocks": 40000, "input_bytes": os.path.getsize(src) + os.path.getsize(os.path.join(d, "untr_profiles", "profiles")),
                                         "gpu_cli_s": g, "cpu_s_on_sample": c, "sample_maf_blocks": 1500, "cpu_kind": "port (Python restatement)",
                                         "identical_on_sample": exp == open(os.path.join(d, "untr_sample_gpu.maf")).read()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    {"gen": gen, "run": run}[sys.argv[1]](sys.argv[2])
