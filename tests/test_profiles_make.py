"""`mugsy_profiles make`.  The reference is OCaml and cannot run here: the oracle
(oracle/make_oracle.py) is a transcription of its source, pinned by a fixture whose expected bytes were derived BY
HAND from the cited lines (tests/golden/make_handmade.*): "restated from source, not executed".
Hand derivation of the fixture, block 0 (8 columns):
  G1 '+' start 0 size 6  -> range (1, 6);   text ACG--TAC -> gaps (4,5)
  G2 '-' start 10 size 7, src 50 -> range (50-10, 50-10-6) = (40, 34); text AC-TTTAC -> gaps (3,3)
  G3 '+' start 3 size 5  -> range (4, 8);   text --GATTA- -> gaps (1,2) (8,8)
  consensus = fold of combine_text over the rows (m_make.ml:35-45), column by column: 1 A,A,- -> A; 2 C,C,- -> C; 3 G,-,G -> G; 4 -,T,A -> T then T vs A -> N; 5 -,T,T -> T;
                    6 T,T,T -> T; 7 A,A,A -> A; 8 C,C,- -> C        => ACGNTTAC
"""
import os
import sys

import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_oracle  # noqa: E402


def test_oracle_matches_hand_derived_fixture():
    prof, fasta = make_oracle.make(open(os.path.join(GOLDEN, "make_handmade.maf")).read(), "x")
    assert prof == open(os.path.join(GOLDEN, "make_handmade.profiles")).read()
    assert fasta == open(os.path.join(GOLDEN, "make_handmade.fasta")).read()


def test_oracle_matches_second_hand_derived_fixture():
    """tests/golden/make_handmade2.*: reverse-strand rows, a column gapped in every row, consensus 'N' cases, three blocks with a
    comment and blank lines between them, a file that ends without a blank line.  Derived by hand from the cited lines:
      block 0 (6 columns, `a score=3 foo`: the prefix "a score=" is all that is looked at, m_profile_stream.ml:23-32)
        A.c 4 4 - 30: of_maf Reverse (m_range.ml:60-65) = (30-4, 30-4-3) = (26, 23); AC--GT -> gaps (3,4)
        B.c 0 4 + 12: (1, 4);  -C-NGT -> gaps (1,1) (3,3)        C.c 2 2 + 9: (3, 4);  ----aN -> gaps (1,4)
        consensus, combine_text folded over the rows (m_make.ml:15-28,35-45):
          AC--GT x -C-NGT: A|- -> A, C|C -> C, -|- -> -, -|N -> N, G|G, T|T          = AC-NGT
          AC-NGT x ----aN: A, C, - (gapped in every row stays '-'), N|- -> N, G|a -> N (both bases, unequal: case counts),
                           T|N -> N                                                    = AC-NNN
      block 1: A.c 0 3 + 30 --TTT -> (1,3), gaps (1,2); a one-row block is its own consensus
      block 2 (after a blank line and a comment, which drop_until_score skips): B.c 5 2 - 12 -> (12-5, 12-5-1) = (7, 6), G-g ->
        gaps (2,2); C.c 0 1 - 9 -> (9, 9), --T -> gaps (1,2); consensus G|- -> G, -|- -> -, g|T -> N = G-N; the stream ends at
        end of file with idx > 0 (m_profile_stream.ml:55-56)."""
    prof, fasta = make_oracle.make(open(os.path.join(GOLDEN, "make_handmade2.maf")).read(), "y")
    assert prof == open(os.path.join(GOLDEN, "make_handmade2.profiles")).read()
    assert fasta == open(os.path.join(GOLDEN, "make_handmade2.fasta")).read()


def test_oracle_matches_third_hand_derived_fixture():
    """tests/golden/make_handmade3.* (basename hm3), derived by hand from the cited lines before the transcription was run on it:
      * `a` without a score does NOT open a block: drop_until_score (m_profile_stream.ml:23-32) looks for the prefix "a score=" and
        drops everything else, so the `a` line, its `s Z.c` row and the blank line vanish and the first SCORED block is _0000;
      * a `#` line between the rows of a block is skipped and does not advance the row index (m_profile_stream.ml:52-53);
      * fields separated by runs of blanks (split_maf filters empty tokens, :16-21);
      * rows on the strand boundary: A.c 0 3 - 10 -> of_maf Reverse (m_range.ml:60-65) = (10-0, 10-0-2) = (10, 8); T-GA -> gaps (2,2)
                                     B.c 7 3 - 10 -> (10-7, 10-7-2) = (3, 1) (the row ends at base 1); TTG- -> gaps (4,4)
      * a row that is all gap: C.c 5 0 + 12 ---- -> Forward (5+1, 5+0) = (6, 5) (start past end: size 0), gaps (1,4), length 4
        consensus of the block, combine_text folded (m_make.ml:15-28,35-45): T-GA x TTG- = TTGA; TTGA x ---- = TTGA
      * a line that is neither blank, `#` nor `a score=` between two blocks is dropped by drop_until_score; `a score=0 x=1` opens
        _0001, whose ONLY row is all gap: D.c 5 0 + 12 ---- -> (6, 5), gaps (1,4); its consensus is ---- itself
      * last block, ended by the end of the file (idx > 0, :55-56): A.c 9 1 + 10 g -> (10, 10); B.c 9 1 - 10 C -> (10-9, 10-9-0) =
        (1, 1); consensus g x C: both bases, unequal -> N."""
    prof, fasta = make_oracle.make(open(os.path.join(GOLDEN, "make_handmade3.maf")).read(), "hm3")
    assert prof == open(os.path.join(GOLDEN, "make_handmade3.profiles")).read()
    assert fasta == open(os.path.join(GOLDEN, "make_handmade3.fasta")).read()


def test_oracle_rejects_what_the_reference_rejects():
    with pytest.raises(ValueError):
        make_oracle.make("a score=1\ns\ttabbed 0 1 + 1 A\n", "x")  # "Unknown line": not prefixed by "s "
    with pytest.raises(ValueError):
        make_oracle.make("a score=1\n", "x")  # "Expected alignment, did not get"
    with pytest.raises(ValueError):
        make_oracle.make("a score=1\ns g 0 1 * 1 A\n", "x")  # Invalid direction
    with pytest.raises(AssertionError):
        make_oracle.make("a score=1\ns g 0 2 + 9 AC\ns h 0 3 + 9 ACG\n", "x")  # combine_text assert


def test_profiles_written_by_oracle_are_read_back_by_the_translate_parser(tmp_path):
    """The format contract between the two stages: what `make` writes is what the translate path's loader reads."""
    import numpy as np
    from paramugsy_amd import synth
    from paramugsy_amd.translate import Workload
    rng = np.random.default_rng(3)
    blocks = synth.gen_side(rng, ["A.c", "B.c"], 5000, 12, mean_cols=120)
    prof, _ = make_oracle.make(synth.side_to_maf_text(blocks), "l")
    assert prof == synth.rows_to_profiles_text(blocks, "l")  # the generator's shortcut equals the stage's output
    (tmp_path / "l").mkdir()
    (tmp_path / "r").mkdir()
    (tmp_path / "l" / "profiles").write_text(prof)
    (tmp_path / "r" / "profiles").write_text("")
    t = Workload.load(str(tmp_path / "l"), str(tmp_path / "r"), []).tables()
    assert len(t.left["start"]) == sum(len(b.rows) for b in blocks)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["handmade", "handmade2", "handmade3", "synthetic", "empty"])
def test_gpu_make_equals_oracle(case, tmp_path):
    import numpy as np
    from paramugsy_amd import capi, synth
    if case == "handmade":
        maf = open(os.path.join(GOLDEN, "make_handmade.maf")).read()
    elif case in ("handmade2", "handmade3"):
        maf = open(os.path.join(GOLDEN, "make_%s.maf" % case)).read()
    elif case == "synthetic":
        rng = np.random.default_rng(11)
        maf = synth.side_to_maf_text(synth.gen_side(rng, ["A.c", "B.c", "C.c"], 60000, 80, mean_cols=500, gap_rate=0.03, edge_gap_prob=0.4))
    else:
        maf = "##maf version=1\n"
    src = tmp_path / "in.maf"
    src.write_text(maf)
    out = tmp_path / "out"
    out.mkdir()
    capi.check(capi.lib().pm_profiles_make(str(src).encode(), str(out).encode(), b"x", 0))
    prof, fasta = make_oracle.make(maf, "x")
    assert (out / "profiles").read_text() == prof
    assert (out / "sequences.fasta").read_text() == fasta
    if case in ("handmade2", "handmade3"):  # and the hand-derived bytes themselves (the fixtures' basenames are y and hm3)
        capi.check(capi.lib().pm_profiles_make(str(src).encode(), str(out).encode(), b"y" if case == "handmade2" else b"hm3", 0))
        assert (out / "profiles").read_text() == open(os.path.join(GOLDEN, "make_%s.profiles" % case)).read()
        assert (out / "sequences.fasta").read_text() == open(os.path.join(GOLDEN, "make_%s.fasta" % case)).read()


@pytest.mark.gpu
def test_gpu_make_cli_and_errors(tmp_path):
    import subprocess
    from paramugsy_amd import capi
    exe = os.path.join(ROOT, "bin", "mugsy_profiles")
    out = tmp_path / "o" / "nested"
    r = subprocess.run([exe, "make", "-in_maf", os.path.join(GOLDEN, "make_handmade.maf"), "-out_dir", str(out), "-basename", "x"],
                       capture_output=True)
    assert r.returncode == 0, r.stderr
    assert (out / "profiles").read_text() == open(os.path.join(GOLDEN, "make_handmade.profiles")).read()
    assert (out / "sequences.fasta").read_text() == open(os.path.join(GOLDEN, "make_handmade.fasta")).read()
    bad = tmp_path / "bad.maf"
    bad.write_text("a score=1\ns g 0 2 + 9 AC\ns h 0 3 + 9 ACG\n")
    rc = capi.lib().pm_profiles_make(str(bad).encode(), str(tmp_path).encode(), b"x", 0)
    assert rc == capi.PM_E_PARSE
    assert subprocess.run([exe, "make", "-in_maf", str(bad)], capture_output=True).returncode != 0
