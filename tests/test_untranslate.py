"""`mugsy_profiles untranslate`.  OCaml reference, cannot run here: the oracle is a transcription of
its source (oracle/untranslate_oracle.py), pinned by tests/golden/untranslate_handmade whose expected bytes were derived
BY HAND ("restated from source, not executed").

Hand derivation (block x.x_0000, 8 columns; rows as in tests/golden/make_handmade.profiles):
  line 1: `s x.x_0000 2 5 + 8 GN-TTT` -> columns (3,7) forward, 5 non-gap characters.
    G1 (1,6) + gaps (4,5) text ACG--TAC: kept gap (4,5); seq(3) = 1+2 = 3, seq(7) = 1+(7-2-1) = 5 -> start 2 size 3;
        row text cols 3..7 = G--TA walked over G N - T T T -> G - - - T A            => s G1.chr 2 3 + 100 G---TA
    G2 (40,34) - gaps (3,3) text AC-TTTAC: gap (3,3) starts the range -> seq(4) = 40-2 = 38, seq(7) = 40-5 = 35;
        reverse row: start = 50-38 = 12, size 4; text cols 3..7 = -TTTA -> - T - T T A  => s G2.chr 12 4 - 50 -T-TTA
    G3 (4,8) + gaps (1,2)(8,8) text --GATTA-: seq(3) = 4+0, seq(7) = 4+4 = 8 -> start 3 size 5; GATTA -> GA-TTA
  line 2: `s x.x_0000 1 4 - 8 AC-GT` -> columns (7,4): reverse overlap, rows clipped to (4,7), strands flip, text is
    reversed and complemented:
    G1: gap (4,5) starts the range -> seq(6) = 4, seq(7) = 5; real (5,4) on '-' -> start 100-5 = 95 size 2;
        cols 4..7 = --TA reversed AT-- walked over A C - G T -> A T - - -, complemented          => TA---
    G2: seq(4) = 38, seq(7) = 35; real (35,38) on '+' -> start 34 size 4; TTTA reversed ATTT -> AT-TT -> TA-AA
    G3: seq(4) = 5, seq(7) = 8; real (8,5) on '-' -> start 20-8 = 12 size 4; ATTA reversed ATTA -> AT-TA -> TA-AT
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_oracle  # noqa: E402
import untranslate_oracle as uo  # noqa: E402

CASE = os.path.join(GOLDEN, "untranslate_handmade")


def test_oracle_matches_hand_derived_fixture():
    got = uo.untranslate([open(os.path.join(CASE, "profiles")).read()], open(os.path.join(CASE, "in.maf")).read())
    assert got == open(os.path.join(CASE, "expected.maf")).read()


def test_oracle_matches_second_hand_derived_fixture():
    """tests/golden/untranslate_handmade2 (profiles = tests/golden/make_handmade2.profiles): reverse-strand rows, a row that is all
    gap in the asked range (dropped), gaps in the line text, one-base ranges (where the reference loses the strand), pass-through
    of comments / blank lines / `a score=` lines, `##maf` dropped.  Rows: P0 A.c (26,23) gaps (3,4) AC--GT; P1 B.c (1,4) gaps
    (1,1)(3,3) -C-NGT; P2 C.c (3,4) gaps (1,4) ----aN; y.y_0002: P4 B.c (7,6) gaps (2,2) G-g; P5 C.c (9,9) gaps (1,2) --T.
    By hand from m_untranslate.ml:38-123 and m_profile.ml:163-264:
      `s y.y_0000 0 6 + 6 ACGTAC` -> columns (1,6) forward: every row comes back as the MAF row `make` read
         (P0: seq(1) = 26, seq(6) = 26-3 = 23, real (26,23) on '-': start 30-26 = 4 size 4 ...).
      `s y.y_0000 1 3 - 6 T-A-C` -> of_maf Reverse (6-1, 6-1-2) = (5,3): reverse overlap, rows clipped to columns (3,5)
         P0: gap (3,4) starts the range -> seq(5) = 26-2 = 24 twice: real = reverse (24,24) = (24,24), Forward: start 23 size 1,
             strand of reverse (26,23) = '+'; text --G reversed G--, over T-A-C: G - - - - , complemented      => C----
         P1: seq(4) = 2, seq(5) = 3 -> real (3,2) Reverse: start 12-3 = 9 size 2, strand of (4,1) = '-'; -NG reversed GN- over
             T-A-C: G - N - -, complemented                                                                    => C-N--
         P2: seq(5) = 3 twice -> real (3,3) is FORWARD by get_direction (s <= e) so start = 3-1 = 2 (a '-' row would want
             9-3 = 6: the reference's one-base quirk), strand of reverse (3,4) = '-'; --a reversed a-- -> a----  => t----
      `s y.y_0000 2 2 + 6 GT` -> columns (3,4): P0 and P2 are all gap there (subset_profile = None, row dropped);
         P1: gap (3,3) starts the range -> seq(4) = 2 twice: start 1 size 1; -N over GT                        => -N
      `s y.y_0002 0 3 + 3 A--CG-` -> columns (1,3): P4: seq(1) = 7, seq(3) = 7-1 = 6, real (7,6) '-': start 12-7 = 5 size 2; G-g
         over A--CG-: G - - - g -; P5: (9,9) reads Forward (one base): start 8 size 1 '+'; --T over A--CG-: - - - - T -
      `s y.y_0002 0 1 - 3 c` -> of_maf Reverse (3,3), which is Forward by get_direction: columns (3,3), no reversal;
         P4: seq(3) = 6: real (6,6) Forward -> start 5, strand of (7,6) = '-'                                  => s B.c 5 1 - 12 g
         P5: seq(3) = 9: start 8 '+'                                                                           => s C.c 8 1 + 9 T"""
    case = os.path.join(GOLDEN, "untranslate_handmade2")
    got = uo.untranslate([open(os.path.join(case, "profiles")).read()], open(os.path.join(case, "in.maf")).read())
    assert got == open(os.path.join(case, "expected.maf")).read()


def test_oracle_matches_third_hand_derived_fixture():
    """tests/golden/untranslate_handmade3 (profiles = tests/golden/make_handmade3.profiles), derived by hand from
    m_untranslate.ml:38-151 and m_profile.ml:163-264 before the transcription was run on it.  Rows of hm3.hm3_0000: A.c (10,8) gaps
    (2,2) T-GA src 10; B.c (3,1) gaps (4,4) TTG- (both on '-', B ending at base 1: the strand boundary); C.c (6,5) gaps (1,4) ----
    (all gap).  hm3.hm3_0001: D.c all gap.  hm3.hm3_0002: A.c (10,10) g; B.c (1,1) C.
      `s hm3.hm3_0000 0 4 + 4 TT-GA` -> columns (1,4) forward:
         A: seq(1) = 10, seq(4) = 10-2 = 8 -> real (10,8) Reverse: start 10-10 = 0, size 3, '-'; T-GA over TT-GA          => T--GA
         B: gap (4,4) ends the range -> seq(3) = 3-2 = 1: real (3,1): start 10-3 = 7, size 3, '-'; TTG- over TT-GA       => TT-G-
         C: its one gap IS the range: subset_profile = None, no line
      the `#` line inside the block passes through where it stands (m_untranslate.ml:131-137)
      `s hm3.hm3_0002 0 1 - 1 g` -> of_maf Reverse (1-0, 1-0-0) = (1,1), Forward by get_direction (s <= e): no reversal, no
         complement; A (10,10): start 9 size 1 '+'; B (1,1) -- a '-' row in the MAF `make` read -- reads Forward: start 0, '+'
      `s hm3.hm3_0001 0 4 + 4 ----` -> D is all gap: the block keeps only its `a score=1` line
      `s hm3.hm3_0000 1 2 - 4 C-A` -> of_maf Reverse (4-1, 4-1-1) = (3,2): columns (2,3), reversed:
         A: gap (2,2) starts the range -> seq(3) = 10-1 = 9 twice; real = reverse (9,9), Forward: start 8 size 1; strand of
            reverse (10,8) = (8,10) = '+'; -G reversed G-, over C-A: G - -, complemented                                  => C--
         B: no gap in (2,3): seq(2) = 2, seq(3) = 1 -> real = reverse (2,1) = (1,2): start 0 size 2 '+'; TG reversed GT over C-A:
            G - T, complemented                                                                                            => C-A
         C: gap (1,4) clipped to (2,3) is the whole range: None."""
    case = os.path.join(GOLDEN, "untranslate_handmade3")
    got = uo.untranslate([open(os.path.join(case, "profiles")).read()], open(os.path.join(case, "in.maf")).read())
    assert got == open(os.path.join(case, "expected.maf")).read()


def synthetic_case(seed):
    """Two profile sets from `make`, and a fake mugsy MAF whose `s` lines cover random column ranges of random blocks on
    either strand, with a few gap columns sprinkled into the line text."""
    from paramugsy_amd import synth
    rng = np.random.default_rng(seed)
    sets = []
    for side, genomes in (("l", ["L0.c", "L1.c", "L2.c"]), ("r", ["R0.c", "R1.c"])):
        blocks = synth.gen_side(rng, genomes, 20000, 25, mean_cols=150, gap_rate=0.04, edge_gap_prob=0.4)
        prof, _ = make_oracle.make(synth.side_to_maf_text(blocks), side)
        sets.append((side, blocks, prof))
    lines = ["##maf version=1 scoring=mugsy", "# produced by a fake mugsyWGA"]
    for _ in range(60):
        lines.append("a score=%d label=1 mult=2" % int(rng.integers(0, 999)))
        for _r in range(int(rng.integers(1, 3))):
            side, blocks, _p = sets[int(rng.integers(0, 2))]
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
    return [s[2] for s in sets], "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed", [1, 2])
def test_oracle_output_is_a_consistent_maf(seed):
    """Size-independent property: every emitted row has as many bases as its size says, and its coordinates lie in the genome."""
    profs, maf = synthetic_case(seed)
    out = uo.untranslate(profs, maf)
    n = 0
    for l in out.split("\n"):
        if l.startswith("s "):
            _, name, start, size, d, src, text = l.split(" ")
            assert len(text) - text.count("-") == int(size)
            assert 0 <= int(start) and int(start) + int(size) <= int(src)
            n += 1
    assert n > 100


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["handmade", "handmade2", "handmade3", "synthetic1", "synthetic2"])
def test_gpu_untranslate_equals_oracle(case, tmp_path):
    import ctypes as C
    from paramugsy_amd import capi
    if case.startswith("handmade"):
        cdir = os.path.join(GOLDEN, "untranslate_" + case)
        profs, maf = [open(os.path.join(cdir, "profiles")).read()], open(os.path.join(cdir, "in.maf")).read()
    else:
        profs, maf = synthetic_case(int(case[-1]))
    dirs = []
    for k, p in enumerate(profs):
        d = tmp_path / ("p%d" % k)
        d.mkdir()
        (d / "profiles").write_text(p)
        dirs.append(str(d).encode())
    (tmp_path / "in.maf").write_text(maf)
    arr = (C.c_char_p * len(dirs))(*dirs)
    capi.check(capi.lib().pm_untranslate(arr, len(dirs), str(tmp_path / "in.maf").encode(), str(tmp_path / "out.maf").encode(), 0))
    assert (tmp_path / "out.maf").read_text() == uo.untranslate(profs, maf)
    if case.startswith("handmade"):  # the hand-derived bytes themselves
        assert (tmp_path / "out.maf").read_text() == open(os.path.join(GOLDEN, "untranslate_" + case, "expected.maf")).read()
    # and through the executable, as the task script calls it
    (tmp_path / "dirs.list").write_text("".join(d.decode() + "\n" for d in dirs))
    r = subprocess.run([os.path.join(ROOT, "bin", "mugsy_profiles"), "untranslate", "-profile_paths_list", str(tmp_path / "dirs.list"),
                        "-in_maf", str(tmp_path / "in.maf"), "-out_maf", str(tmp_path / "cli.maf")], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "cli.maf").read_text() == (tmp_path / "out.maf").read_text()


@pytest.mark.gpu
def test_gpu_untranslate_failures(tmp_path):
    import ctypes as C
    from paramugsy_amd import capi
    d = tmp_path / "p"
    d.mkdir()
    (d / "profiles").write_text(open(os.path.join(CASE, "profiles")).read())
    arr = (C.c_char_p * 1)(str(d).encode())
    # a range past the block's columns: Profile_idx_out_of_range in the reference
    (tmp_path / "a.maf").write_text("a score=1\ns x.x_0000 5 9 + 20 ACGTACGTA\n")
    rc = capi.lib().pm_untranslate(arr, 1, str(tmp_path / "a.maf").encode(), str(tmp_path / "o").encode(), 0)
    assert rc == capi.PM_E_UNIT
    # an unknown block: Not_found
    (tmp_path / "b.maf").write_text("a score=1\ns nope 0 1 + 8 A\n")
    rc = capi.lib().pm_untranslate(arr, 1, str(tmp_path / "b.maf").encode(), str(tmp_path / "o").encode(), 0)
    assert rc == capi.PM_E_PARSE
    with pytest.raises(Exception):
        uo.untranslate([open(os.path.join(CASE, "profiles")).read()], "a score=1\ns x.x_0000 5 9 + 20 ACGTACGTA\n")
