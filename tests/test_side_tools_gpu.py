"""m_sort_delta and maf_analyzer on the GPU against the bytes the upstream binaries printed (tests/golden) and against
the oracle on fresh inputs, through the C ABI and the drop-in executables."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from paramugsy_amd import capi, synth

pytestmark = pytest.mark.gpu


def run_sort(src, out):
    capi.check(capi.lib().pm_sort_delta(src.encode(), out.encode(), 0))
    return open(out, "rb").read()


def run_maf(src, out):
    capi.check(capi.lib().pm_maf_analyzer(src.encode(), out.encode(), 0))
    return open(out, "rb").read()


@pytest.mark.parametrize("name", ["mixed", "headers_only"])
def test_sort_delta_bytes_equal_reference_golden(name, tmp_path):
    got = run_sort(os.path.join(GOLDEN, "sort_%s.delta" % name), str(tmp_path / "o"))
    assert got == open(os.path.join(GOLDEN, "sort_%s.expected" % name), "rb").read()


@pytest.mark.parametrize("name", ["synthetic", "adjacent_shuffled", "highly_stitchable"])
def test_maf_analyzer_bytes_equal_reference_golden(name, tmp_path):
    src = os.path.join(GOLDEN, "highly_stitchable.maf" if name == "highly_stitchable" else "maf_%s.maf" % name)
    got = run_maf(src, str(tmp_path / "o"))
    assert got == open(os.path.join(GOLDEN, "maf_%s.expected" % name), "rb").read()


def test_drop_in_executables(tmp_path):
    exp = open(os.path.join(GOLDEN, "sort_mixed.expected"), "rb").read()
    with open(os.path.join(GOLDEN, "sort_mixed.delta"), "rb") as f:
        r = subprocess.run([os.path.join(ROOT, "bin", "m_sort_delta")], stdin=f, capture_output=True)
    assert r.returncode == 0 and r.stdout == exp
    r = subprocess.run([os.path.join(ROOT, "bin", "maf_analyzer"), os.path.join(GOLDEN, "highly_stitchable.maf")], capture_output=True)
    assert r.returncode == 0 and r.stdout == open(os.path.join(GOLDEN, "maf_highly_stitchable.expected"), "rb").read()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sort_delta_equals_oracle_on_fresh_input(seed, oracle_build, tmp_path):
    rng = np.random.default_rng(seed)
    text = synth.gen_delta_text(rng, ["b", "a", "c", "ab"], ["y", "x"], 200000, 200000, 2000, mean_len=600, group=2)
    src = tmp_path / "in.delta"
    src.write_text(text)
    with open(src, "rb") as f:
        exp = subprocess.run([os.path.join(oracle_build, "oracle_m_sort_delta")], stdin=f, capture_output=True, check=True).stdout
    assert run_sort(str(src), str(tmp_path / "o")) == exp


@pytest.mark.parametrize("seed", [1, 2])
def test_maf_analyzer_equals_oracle_with_overlapping_rows(seed, oracle_build, tmp_path):
    """Overlapping rows make the upstream insertion order dependent: the exact replay kernel is used."""
    rng = np.random.default_rng(seed)
    blocks = synth.gen_side(rng, ["A", "B", "C"], 4000, 40, mean_cols=80, spacing=3, gap_rate=0.0, edge_gap_prob=0.0)
    more = synth.gen_side(rng, ["A", "B"], 4000, 25, mean_cols=90, spacing=20, gap_rate=0.0, edge_gap_prob=0.0)  # overlaps the first set
    order = rng.permutation(len(blocks) + len(more))
    allb = blocks + more
    src = tmp_path / "in.maf"
    src.write_text(synth.side_to_maf_text([allb[i] for i in order]))
    exp = subprocess.run([os.path.join(oracle_build, "oracle_maf_analyzer"), str(src)], capture_output=True, check=True).stdout
    assert run_maf(str(src), str(tmp_path / "o")) == exp
    assert exp.count(b"\n") > 10


def test_maf_edge_cases_equal_oracle(oracle_build, tmp_path):
    cases = {
        "no_trailing_newline": "a score=0 x\ns A 0 10 + 100 ACGTACGTAC\n\na score=0 y\ns A 20 5 + 100 ACGTA",
        "last_block_on_unterminated_a_line": "a score=0 x\ns A 0 10 + 100 ACGTACGTAC\n\na score=0 y",
        "comments_and_reverse": "##maf version=1\n# c\n\na score=0 x\ns B 0 10 - 50 ACGTACGTAC\ns A 5 5 + 30 ACGTA\n\n",
        "empty": "",
        "blocks_without_blank_line": "a score=0 x\ns A 0 10 + 100 ACGTACGTAC\na score=0 y\ns A 50 5 + 100 ACGTA\n",
    }
    for name, text in cases.items():
        src = tmp_path / (name + ".maf")
        src.write_text(text)
        exp = subprocess.run([os.path.join(oracle_build, "oracle_maf_analyzer"), str(src)], capture_output=True, check=True).stdout
        assert run_maf(str(src), str(tmp_path / "o")) == exp, name
