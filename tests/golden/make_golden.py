#!/usr/bin/env python3
"""Regenerates the golden fixtures under tests/golden/ from the upstream reference ITSELF.

Needs oracle/_ref/ (the reference's own C++ compiled from /root/reference by `make -C oracle ref`), so it
only runs in the dev container; the GPU box and CI use the committed outputs.  Fixtures are data only:
seeded synthetic inputs (paramugsy_amd/synth.py) and the bytes the reference binaries printed for them.

  translate_<name>/   profiles-l/profiles, profiles-r/profiles, nucmer_*.delta, nucmer.list  (inputs)
                      expected.delta   = oracle/_ref/m_translate profiles-l profiles-r nucmer.list expected.delta
                                         (run with cwd = the case directory so the first line is stable)
  sort_<name>.delta / sort_<name>.expected      = oracle/_ref/m_sort_delta < in > expected
  maf_<name>.maf / maf_<name>.expected          = oracle/_ref/maf_analyzer in > expected
  units_cmds.txt / units_expected.txt           = oracle/_ref/ref_units < cmds > expected
  stage_pin/          what the reference's compiled C++ says about the inputs and outputs of the two OCaml stages around the path
                      (round 4; tests/test_stage_pins.py): side_l.maf, side_r.maf (seeded synthetic sides), l/profiles, r/profiles
                      (what `mugsy_profiles make` writes for them -- oracle/make_oracle.py's bytes, the HIP path must print the same),
                      in.maf (a fake mugsyWGA output over column ranges of those profiles), cmds.txt / expected.txt =
                      oracle/_ref/ref_units < cmds > expected (run with cwd = tests/golden): Maf_read_stream and of_maf over the
                      sides, read_profile_file over the profiles, profile_idx_of_seq_idx for the bases of every row,
                      subset_profile and seq_idx_of_profile_idx for every (row profile, column range) untranslate asks for
"""
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from paramugsy_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")

TRANSLATE_CASES = {
    # name: (seed, kwargs)
    "typical": (101, dict(n_left=2, n_right=2, genome_len=20000, n_blocks=25, entries_per_delta=25, n_deltas=2)),
    "gappy": (202, dict(n_left=2, n_right=3, genome_len=12000, n_blocks=30, mean_cols=200, gap_rate=0.05, mean_gap=6.0,
                        indel_rate=0.02, mean_indel=4.0, adjacent_prob=0.1, edge_gap_prob=0.5, entries_per_delta=30, n_deltas=1)),
    "reverse": (303, dict(n_left=2, n_right=2, genome_len=8000, n_blocks=30, mean_cols=120, gap_rate=0.08, indel_rate=0.05,
                          mean_len=400, entries_per_delta=40, rev_prob=0.5, delta_rev_prob=0.5, spacing=5, n_deltas=1)),
    "tiny_blocks": (1007, dict(n_left=2, n_right=2, genome_len=3000, n_blocks=150, mean_cols=8, gap_rate=0.1, mean_gap=3.0,
                               indel_rate=0.05, mean_indel=8.0, mean_len=150, entries_per_delta=40, spacing=3,
                               edge_gap_prob=0.4, adjacent_prob=0.15, delta_rev_prob=0.4, n_deltas=1)),
}


def relativise(case_dir: str, w: synth.Workload) -> None:
    """nucmer.list must hold paths relative to the case directory (the tests run with cwd = case dir)."""
    with open(w.list_path, "w") as f:
        for p in w.delta_paths:
            f.write(os.path.basename(p) + "\n")


def make_translate_cases() -> None:
    for name, (seed, kw) in TRANSLATE_CASES.items():
        case = os.path.join(HERE, "translate_" + name)
        shutil.rmtree(case, ignore_errors=True)
        w = synth.make_workload(case, seed, **kw)
        relativise(case, w)
        r = subprocess.run([os.path.join(REF, "m_translate"), "profiles-l", "profiles-r", "nucmer.list", "expected.delta"], cwd=case)
        assert r.returncode == 0, (name, r.returncode)
        print("translate_%s: %d bytes expected" % (name, os.path.getsize(os.path.join(case, "expected.delta"))))
    # empty / degenerate job: no delta files at all, and a side with no rows
    case = os.path.join(HERE, "translate_empty")
    shutil.rmtree(case, ignore_errors=True)
    os.makedirs(os.path.join(case, "profiles-l"))
    os.makedirs(os.path.join(case, "profiles-r"))
    open(os.path.join(case, "profiles-l", "profiles"), "w").close()
    rng = np.random.default_rng(7)
    synth.write_side(os.path.join(case, "profiles-r"), synth.gen_side(rng, ["R0.chr"], 2000, 3), "r")
    with open(os.path.join(case, "nucmer_0.delta"), "w") as f:
        f.write(synth.gen_delta_text(rng, ["L0.chr"], ["R0.chr"], 2000, 2000, 5, mean_len=300))
    with open(os.path.join(case, "nucmer.list"), "w") as f:
        f.write("nucmer_0.delta\n")
    r = subprocess.run([os.path.join(REF, "m_translate"), "profiles-l", "profiles-r", "nucmer.list", "expected.delta"], cwd=case)
    assert r.returncode == 0
    print("translate_empty: %d bytes expected" % os.path.getsize(os.path.join(case, "expected.delta")))


def make_sort_cases() -> None:
    rng = np.random.default_rng(404)
    text = synth.gen_delta_text(rng, ["b.chr", "a.chr", "c.chr"], ["y.chr", "x.chr"], 50000, 50000, 60, mean_len=800, group=3)
    # the reference's one in-tree known-answer vector (lib/profiles_lib/m_delta.cc:43-49) as an extra entry
    text += ">a.chr x.chr 50000 50000\n1 2000 1 2000 0 0 0\n106\n-6\n1797\n-9\n-9\n-1\n7\n1\n0\n"
    for name, body in (("mixed", text), ("headers_only", "/s/ref /s/qry\nNUCMER\n")):
        src = os.path.join(HERE, "sort_%s.delta" % name)
        with open(src, "w") as f:
            f.write(body)
        with open(src) as fin, open(os.path.join(HERE, "sort_%s.expected" % name), "w") as fout:
            r = subprocess.run([os.path.join(REF, "m_sort_delta")], stdin=fin, stdout=fout)
        assert r.returncode == 0
        print("sort_%s ok" % name)


def make_maf_cases() -> None:
    rng = np.random.default_rng(505)
    blocks = synth.gen_side(rng, ["A", "B", "C"], 5000, 20, mean_cols=150, spacing=30)
    cases = {"synthetic": synth.side_to_maf_text(blocks)}
    # adjacent blocks (spacing 0..1) so that the merge branches of _insert are taken, shuffled block order
    blocks2 = synth.gen_side(rng, ["A", "B"], 3000, 25, mean_cols=60, spacing=2, gap_rate=0.0, edge_gap_prob=0.0)
    order = rng.permutation(len(blocks2))
    cases["adjacent_shuffled"] = synth.side_to_maf_text([blocks2[i] for i in order])
    for name, body in cases.items():
        src = os.path.join(HERE, "maf_%s.maf" % name)
        with open(src, "w") as f:
            f.write(body)
        with open(os.path.join(HERE, "maf_%s.expected" % name), "w") as fout:
            r = subprocess.run([os.path.join(REF, "maf_analyzer"), src], stdout=fout)
        assert r.returncode == 0
        print("maf_%s ok" % name)
    # BASELINE config 1: the reference's own tests/highly_stitchable.maf.  That file is a test vector the reference holds
    # (data, not source); tests/golden/highly_stitchable.maf is a byte copy of it so that the GPU box, which has no
    # /root/reference, can run the case.  The expected output is what the reference binary printed for it here.
    ref_maf = "/root/reference/tests/highly_stitchable.maf"
    if os.path.exists(ref_maf):
        with open(os.path.join(HERE, "maf_highly_stitchable.expected"), "w") as fout:
            subprocess.run([os.path.join(REF, "maf_analyzer"), ref_maf], stdout=fout, check=True)


def make_unit_cases() -> None:
    rng = np.random.default_rng(606)
    lines = ["dparse 1 2000 1 2000 106 -6 1797 -9 -9 -1 7 1", "d2o", "drev"]  # m_delta.cc:43-49
    for _ in range(120):
        size = int(rng.integers(1, 60))
        text = ["A"] * size
        for _g in range(int(rng.integers(0, 6))):
            at = int(rng.integers(0, len(text) + 1))
            text[at:at] = ["-"] * int(rng.integers(1, 5))
        t = "".join(text).encode()
        gs = synth.gaps_of_text(t)
        start = int(rng.integers(1, 1000))
        fwd = rng.random() < 0.6
        s, e = (start, start + size - 1) if fwd else (start + size - 1, start)
        plen = len(t) + (int(rng.integers(-3, 4)) if rng.random() < 0.1 else 0)
        lines.append("profile %d %d %d %d %s" % (s, e, plen, len(gs), " ".join("%d %d" % g for g in gs)))
        lo, hi = min(s, e), max(s, e)
        for _q in range(10):
            k = int(rng.integers(0, 4))
            if k == 0:
                lines.append("p2s %d" % int(rng.integers(lo - 2, hi + 3)))
            elif k == 1:
                lines.append("s2p %d" % int(rng.integers(-1, plen + 3)))
            elif k == 2:
                lines.append("sub %d %d" % (int(rng.integers(-1, plen + 3)), int(rng.integers(-1, plen + 3))))
            else:
                lines.append("subseq %d %d" % (int(rng.integers(lo - 1, hi + 2)), int(rng.integers(lo - 1, hi + 2))))
        offs = synth._delta_offsets(rng, int(rng.integers(5, 80)), 0.1, 2.0, 0.1)[0]
        lines.append("dparse 10 100 200 300 " + " ".join(str(v) for v in offs))
        lines.append("drev")
        lines.append("d2o")
    cmds = "\n".join(lines) + "\n"
    with open(os.path.join(HERE, "units_cmds.txt"), "w") as f:
        f.write(cmds)
    r = subprocess.run([os.path.join(REF, "ref_units")], input=cmds.encode(), capture_output=True, check=True)
    with open(os.path.join(HERE, "units_expected.txt"), "wb") as f:
        f.write(r.stdout)
    print("units: %d commands" % len(lines))


def stage_pin_commands(case: str) -> str:
    """The command list of the stage pin (paths relative to tests/golden).  Deterministic in the fixture's files, so that the test can
    rebuild it and hold the committed cmds.txt to it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import untranslate_oracle as uo
    rel = os.path.relpath(case, HERE)
    lines = []
    records = {}
    for side in ("l", "r"):
        lines.append("mafread %s/side_%s.maf" % (rel, side))
        lines.append("profread %s/%s/profiles 0" % (rel, side))
        profs = uo.read_profiles(open(os.path.join(case, side, "profiles")).read())
        records[side] = profs
        for k, p in enumerate(profs):
            lines.append("profpick %s/%s/profiles %d" % (rel, side, k))
            fwd = p.range[0] <= p.range[1]
            n_bases = abs(p.range[0] - p.range[1]) + 1 if p.length - sum(b - a + 1 for a, b in p.gaps) > 0 else 0
            step = max(1, n_bases // 12)
            for j in list(range(0, n_bases, step)) + ([n_bases - 1] if n_bases else []):
                lines.append("p2s %d" % (p.range[0] + j if fwd else p.range[0] - j))  # the j-th base of the row
    # every (row profile, column range) the untranslate of in.maf asks for
    by_block = {}
    for side in ("l", "r"):
        for k, p in enumerate(records[side]):
            by_block.setdefault(p.major, []).append((side, k, p))
    for l in open(os.path.join(case, "in.maf")).read().split("\n"):
        if not l.startswith("s "):
            continue
        _, name, start, size, d, src_size, _text = [t for t in l.split(" ") if t != ""]
        lines.append("ofmaf %s %s %s %s" % (start, size, src_size, d))
        ov = uo.of_maf(int(start), int(size), int(src_size), d)
        s_, e_ = min(ov), max(ov)
        for side, k, p in by_block[name]:
            lines.append("profpick %s/%s/profiles %d" % (rel, side, k))
            lines.append("sub %d %d" % (s_, e_))
            lines.append("s2p %d" % s_)
            lines.append("s2p %d" % e_)
    return "\n".join(lines) + "\n"


def make_stage_pin() -> None:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import make_oracle
    case = os.path.join(HERE, "stage_pin")
    shutil.rmtree(case, ignore_errors=True)
    os.makedirs(case)
    rng = np.random.default_rng(20261004)
    sides = {}
    for side, genomes in (("l", ["L0.c", "L1.c", "L2.c"]), ("r", ["R0.c", "R1.c"])):
        blocks = synth.gen_side(rng, genomes, 6000, 14, mean_cols=70, gap_rate=0.06, mean_gap=3.0, edge_gap_prob=0.5, rev_prob=0.4)
        maf = synth.side_to_maf_text(blocks)
        with open(os.path.join(case, "side_%s.maf" % side), "w") as f:
            f.write(maf)
        prof, fasta = make_oracle.make(maf, side)
        os.makedirs(os.path.join(case, side))
        with open(os.path.join(case, side, "profiles"), "w") as f:
            f.write(prof)
        with open(os.path.join(case, side, "sequences.fasta"), "w") as f:
            f.write(fasta)
        sides[side] = blocks
    lines = ["##maf version=1 scoring=mugsy", "# a fake mugsyWGA output over column ranges of the two sides' profiles"]
    for _ in range(40):
        lines.append("a score=%d label=1 mult=2" % int(rng.integers(0, 999)))
        for _r in range(int(rng.integers(1, 4))):
            side = "lr"[int(rng.integers(0, 2))]
            blocks = sides[side]
            b = int(rng.integers(0, len(blocks)))
            cols = len(blocks[b].rows[0].text)
            size = int(rng.integers(1, cols + 1))
            start = int(rng.integers(0, cols - size + 1))
            text = list("ACGT"[int(x)] for x in rng.integers(0, 4, size=size))
            for _g in range(int(rng.integers(0, 3))):
                text.insert(int(rng.integers(0, len(text) + 1)), "-")
            lines.append("s %s.%s_%04d %d %d %s %d %s" % (side, side, b, start, size, "+" if rng.random() < 0.6 else "-", cols, "".join(text)))
        lines.append("")
    with open(os.path.join(case, "in.maf"), "w") as f:
        f.write("\n".join(lines) + "\n")
    cmds = stage_pin_commands(case)
    with open(os.path.join(case, "cmds.txt"), "w") as f:
        f.write(cmds)
    r = subprocess.run([os.path.join(REF, "ref_units")], input=cmds.encode(), capture_output=True, check=True, cwd=HERE)
    with open(os.path.join(case, "expected.txt"), "wb") as f:
        f.write(r.stdout)
    print("stage_pin: %d commands, %d bytes expected" % (cmds.count("\n"), len(r.stdout)))


if __name__ == "__main__":
    if not os.path.exists(os.path.join(REF, "m_translate")):
        sys.exit("oracle/_ref missing: run `make -C oracle ref` where /root/reference exists")
    if len(sys.argv) > 1 and sys.argv[1] == "stage_pin":
        make_stage_pin()
        sys.exit(0)
    make_translate_cases()
    make_sort_cases()
    make_maf_cases()
    make_unit_cases()
    make_stage_pin()
