"""What the reference's COMPILED C++ says about the two OCaml stages around the path (SURVEY.md 8f.1 `mugsy_profiles make`, 8f.2
`untranslate`).  The OCaml cannot run here, so oracle/make_oracle.py and oracle/untranslate_oracle.py are
transcriptions, "restated from source, not executed" -- but several pieces of them have C++ twins in lib/profiles_lib that DO run
here (oracle/_ref/ref_units = oracle/ref_units_driver.cc linked against the reference's own objects):

  of_maf                     m_range.hh:106-115        the range of an `s` line                      (OCaml twin m_range.ml:60-65)
  Maf_read_stream::next      maf_read_stream.cc:7-45   blocks and rows of a MAF file                 (m_profile_stream.ml:16-58)
  read_profile_file          m_profile.cc:15-85        the records of a `profiles` file              (m_profile.ml:69-120, 122-135)
  profile_idx_of_seq_idx     m_profile.cc:91-112       where a base of a row sits among its columns  (m_profile.ml:146-161)
  subset_profile             m_profile.cc:160-206      a row profile cut to a column range           (m_profile.ml:189-239)
  seq_idx_of_profile_idx     m_profile.cc:114-149      the base at a column                          (m_profile.ml:163-181)

tests/golden/stage_pin/ holds two seeded MAF sides, the `profiles` files `make` writes for them, a fake mugsyWGA output over column
ranges of those profiles, and expected.txt = what oracle/_ref/ref_units printed for cmds.txt (tests/golden/make_golden.py stage_pin).
EXECUTED-PINNED by these tests (lines of the transcriptions that now rest on reference code that ran):
  make_oracle.py        the row header -- seq name, range (of_maf), src_size -- of every record; the record LAYOUT (the reference's
                        reader parses the file back to the same numbers and text); the GAP LIST of every row: the reference's own
                        profile_idx_of_seq_idx, run on the gap list `make` wrote, puts every sampled base of the row at the column where
                        the row's text has it;
  untranslate_oracle.py of_maf of every `s` line; subset_profile's range and gap list and seq_idx_of_profile_idx, for every (row
                        profile, column range) the stage asks for -- except p_length, where the OCaml differs by design of its authors
                        (m_profile.ml:232: |s - e|; C++ m_profile.hh:54-63: |range| + gap columns), asserted as the quirk it is.
STILL HAND-PINNED ONLY (no C++ twin exists): combine_text / the consensus FASTA (m_make.ml:15-62), expand_text and the reverse
complement (m_untranslate.ml:15-52, 71-123), drop_until_score's treatment of unscored `a` lines and `#` lines inside blocks
(m_profile_stream.ml:23-32, 52-53; the C++ reader differs there, which is why the pinned sides are plain MAF)."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, GOLDEN)
import make_oracle  # noqa: E402
import untranslate_oracle as uo  # noqa: E402

CASE = os.path.join(GOLDEN, "stage_pin")


def replies():
    """[(command, [reply lines])] of the committed reference run."""
    cmds = open(os.path.join(CASE, "cmds.txt")).read().splitlines()
    out = open(os.path.join(CASE, "expected.txt")).read().splitlines()
    res, at = [], 0
    for c in cmds:
        if c.startswith("mafread") or c.startswith("profread"):
            end = out.index("END", at)
            res.append((c, out[at:end]))
            at = end + 1
        else:
            res.append((c, [out[at]]))
            at += 1
    assert at == len(out)
    return res


def test_the_committed_commands_are_what_the_fixture_implies():
    import make_golden
    assert make_golden.stage_pin_commands(CASE) == open(os.path.join(CASE, "cmds.txt")).read()
    for side in ("l", "r"):  # and the profiles files are what the transcription of `make` writes for the sides
        prof, fasta = make_oracle.make(open(os.path.join(CASE, "side_%s.maf" % side)).read(), side)
        assert prof == open(os.path.join(CASE, side, "profiles")).read()
        assert fasta == open(os.path.join(CASE, side, "sequences.fasta")).read()


def test_this_repos_restatement_prints_the_references_lines(oracle_build):
    """oracle/pm_oracle.cc (range_of_maf, read_maf_block, read_profile, the index conversions, subset_profile) through the twin of the
    driver: byte for byte what the reference printed."""
    r = subprocess.run([os.path.join(ROOT, "oracle", "_build", "oracle_units")], input=open(os.path.join(CASE, "cmds.txt"), "rb").read(),
                       capture_output=True, check=True, cwd=GOLDEN)
    assert r.stdout == open(os.path.join(CASE, "expected.txt"), "rb").read()


def test_the_reference_still_prints_the_committed_lines():
    ref = os.path.join(ROOT, "oracle", "_ref", "ref_units")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/ref_units is built where /root/reference exists")
    r = subprocess.run([ref], input=open(os.path.join(CASE, "cmds.txt"), "rb").read(), capture_output=True, check=True, cwd=GOLDEN)
    assert r.stdout == open(os.path.join(CASE, "expected.txt"), "rb").read()


def test_what_make_writes_is_what_the_reference_reads_and_means():
    rep = replies()
    at = 0
    for side in ("l", "r"):
        (c_maf, maf_lines), (c_prof, prof_lines) = rep[at], rep[at + 1]
        assert c_maf.startswith("mafread") and c_prof.startswith("profread")
        at += 2
        rows = [ln.split(" ") for ln in maf_lines if ln.startswith("ALN ")]
        recs = [ln.split(" ") for ln in prof_lines]
        assert len(rows) == len(recs) > 15
        blocks = [ln for ln in maf_lines if ln.startswith("ENTRY ")]
        k = 0
        for b, entry in enumerate(blocks):
            for r in range(int(entry.split(" ")[3])):
                _, genome, start, size, src_size, rs, re, text = rows[k]
                _, major, minor, seq, ps, pe, length, psrc, ngaps = recs[k][:9]
                gaps = [(int(recs[k][9 + 2 * g]), int(recs[k][10 + 2 * g])) for g in range(int(ngaps))]
                ptext = recs[k][9 + 2 * int(ngaps)]
                # the record's header is the reference's reading of the `s` line; names as m_profile_stream.ml:40,65 makes them
                assert (major, minor, seq) == ("%s.%s_%04d" % (side, side, b), str(r), genome)
                assert (ps, pe) == (rs, re) and psrc == src_size and int(length) == len(text) and ptext == text
                # the gap list means what the reference takes it to mean: its own profile_idx_of_seq_idx, on the gap list `make`
                # wrote, finds every sampled base of the row at the column where the text has it
                cols = [i + 1 for i, ch in enumerate(text) if ch != "-"]
                assert rep[at][0].startswith("profpick") and rep[at][1][0].split(" ")[1:4] == [major, minor, seq]
                at += 1
                fwd = int(rs) <= int(re)
                while at < len(rep) and rep[at][0].startswith("p2s"):
                    si = int(rep[at][0].split(" ")[1])
                    j = si - int(rs) if fwd else int(rs) - si
                    assert rep[at][1][0] == "IDX %d" % cols[j], (major, minor, si)
                    at += 1
                # (and the runs of '-' of the text are the gap list: what gaps_of_text, m_profile.ml:29-47, is restated as)
                runs, i = [], 0
                while i < len(text):
                    if text[i] == "-":
                        j = i
                        while j + 1 < len(text) and text[j + 1] == "-":
                            j += 1
                        runs.append((i + 1, j + 1))
                        i = j + 1
                    else:
                        i += 1
                assert gaps == runs
                k += 1
        assert k == len(rows)
    assert rep[at][0].startswith("ofmaf")


def test_untranslates_index_arithmetic_is_the_references():
    rep = replies()
    at = next(i for i, (c, _) in enumerate(rep) if c.startswith("ofmaf"))
    profs = {side: uo.read_profiles(open(os.path.join(CASE, side, "profiles")).read()) for side in ("l", "r")}
    n_sub = n_none = 0
    for l in open(os.path.join(CASE, "in.maf")).read().split("\n"):
        if not l.startswith("s "):
            continue
        _, name, start, size, d, src_size, _text = [t for t in l.split(" ") if t != ""]
        ov = uo.of_maf(int(start), int(size), int(src_size), d)
        assert rep[at][1][0] == "RANGE %d %d" % ov
        at += 1
        s, e = min(ov), max(ov)
        while at < len(rep) and rep[at][0].startswith("profpick"):
            side, k = rep[at][0].split(" ")[1].split("/")[1], int(rep[at][0].split(" ")[2])
            p = profs[side][k]
            assert p.major == name
            (c_sub, (r_sub,)), (_, (r_s,)), (_, (r_e,)) = rep[at + 1], rep[at + 2], rep[at + 3]
            assert c_sub == "sub %d %d" % (s, e)
            at += 4
            for pi, reply in ((s, r_s), (e, r_e)):
                v = uo.seq_idx_of_profile_idx(p, pi)
                assert reply == ("NONE" if v is None else "IDX %d" % v)
            mine = uo.subset_profile(p, s, e)
            if r_sub == "NONE":
                assert mine is None
                n_none += 1
                continue
            f = r_sub.split(" ")
            assert f[0] == "PROFILE" and mine is not None
            gaps = [(int(f[5 + 2 * g]), int(f[6 + 2 * g])) for g in range(int(f[4]))]
            assert (int(f[1]), int(f[2])) == mine.range and gaps == mine.gaps
            # the one place the twins differ by design: C++ recomputes p_length (m_profile.hh:54-63), OCaml keeps |s - e| (m_profile.ml:232)
            assert int(f[3]) == abs(mine.range[0] - mine.range[1]) + 1 + sum(b - a + 1 for a, b in gaps)
            assert mine.length == abs(s - e)
            n_sub += 1
    assert at == len(rep) and n_sub > 100 and n_none >= 1


@pytest.mark.gpu
def test_the_hip_stages_print_the_pinned_bytes(tmp_path):
    """pm_profiles_make on the pinned sides writes the pinned `profiles` files; pm_untranslate over them prints what the transcription
    -- whose index arithmetic the tests above hold to the reference's -- prints for the pinned mugsy MAF."""
    import ctypes as C
    from paramugsy_amd import capi
    dirs = []
    for side in ("l", "r"):
        out = str(tmp_path / side)
        os.makedirs(out)
        capi.check(capi.lib().pm_profiles_make(os.path.join(CASE, "side_%s.maf" % side).encode(), out.encode(), side.encode(), 0))
        assert open(os.path.join(out, "profiles")).read() == open(os.path.join(CASE, side, "profiles")).read()
        assert open(os.path.join(out, "sequences.fasta")).read() == open(os.path.join(CASE, side, "sequences.fasta")).read()
        dirs.append(out.encode())
    arr = (C.c_char_p * 2)(*dirs)
    got = str(tmp_path / "out.maf")
    capi.check(capi.lib().pm_untranslate(arr, 2, os.path.join(CASE, "in.maf").encode(), got.encode(), 0))
    want = uo.untranslate([open(os.path.join(CASE, side, "profiles")).read() for side in ("l", "r")], open(os.path.join(CASE, "in.maf")).read())
    assert open(got).read() == want
    assert want.count("\ns ") > 100
