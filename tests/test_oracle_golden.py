"""The CPU oracle against the golden fixtures the upstream reference itself produced
(tests/golden/make_golden.py), plus the reference's one in-tree known-answer vector.  CPU only."""
import filecmp
import os
import subprocess

import pytest

from conftest import GOLDEN

TRANSLATE_CASES = ["typical", "gappy", "reverse", "tiny_blocks", "empty"]


@pytest.mark.parametrize("name", TRANSLATE_CASES)
def test_oracle_m_translate_matches_reference_bytes(name, oracle_build, tmp_path):
    case = os.path.join(GOLDEN, "translate_" + name)
    out = tmp_path / "out.delta"
    r = subprocess.run([os.path.join(oracle_build, "oracle_m_translate"), "profiles-l", "profiles-r", "nucmer.list", str(out)], cwd=case)
    assert r.returncode == 0
    assert filecmp.cmp(str(out), os.path.join(case, "expected.delta"), shallow=False)


@pytest.mark.parametrize("name", ["mixed", "headers_only"])
def test_oracle_m_sort_delta_matches_reference_bytes(name, oracle_build):
    with open(os.path.join(GOLDEN, "sort_%s.delta" % name), "rb") as f:
        r = subprocess.run([os.path.join(oracle_build, "oracle_m_sort_delta")], stdin=f, capture_output=True)
    assert r.returncode == 0
    assert r.stdout == open(os.path.join(GOLDEN, "sort_%s.expected" % name), "rb").read()


@pytest.mark.parametrize("name", ["synthetic", "adjacent_shuffled", "highly_stitchable"])
def test_oracle_maf_analyzer_matches_reference_bytes(name, oracle_build):
    # highly_stitchable.maf is the reference's own test data file (BASELINE config 1)
    src = os.path.join(GOLDEN, "highly_stitchable.maf" if name == "highly_stitchable" else "maf_%s.maf" % name)
    r = subprocess.run([os.path.join(oracle_build, "oracle_maf_analyzer"), src], capture_output=True)
    assert r.returncode == 0
    assert r.stdout == open(os.path.join(GOLDEN, "maf_%s.expected" % name), "rb").read()


def test_highly_stitchable_expected_text():
    # four groups, each "<G>\t81\t100"
    exp = open(os.path.join(GOLDEN, "maf_highly_stitchable.expected")).read()
    assert exp == "".join("--------\n%s\t81\t100\n" % g for g in "ABCD")


def test_oracle_unit_functions_match_reference(oracle_build):
    cmds = open(os.path.join(GOLDEN, "units_cmds.txt"), "rb").read()
    r = subprocess.run([os.path.join(oracle_build, "oracle_units")], input=cmds, capture_output=True)
    assert r.returncode == 0
    assert r.stdout == open(os.path.join(GOLDEN, "units_expected.txt"), "rb").read()


def test_known_answer_vector_from_reference_comment(oracle_build):
    # lib/profiles_lib/m_delta.cc:43-49: 106 -6 1797 -9 -9 -1 7 1
    #   ref gaps (112,112) (1918,1918) (1927,1928); query gaps (106,106) (1909,1909) (1935,1936)
    r = subprocess.run([os.path.join(oracle_build, "oracle_units")], input=b"dparse 1 2000 1 2000 106 -6 1797 -9 -9 -1 7 1\nd2o\n",
                       capture_output=True)
    lines = r.stdout.decode().splitlines()
    assert lines[0] == "DELTA 1 2000 1 2000 3 112 112 1918 1918 1927 1928 3 106 106 1909 1909 1935 1936"
    assert lines[1] == "OFFSETS 106 -6 1797 -9 -9 -1 7 1 0"
