"""Host logic of the product (file parsers, per-sequence index, unit enumeration) against the oracle, on CPU.
pm_workload_* needs no device."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from paramugsy_amd import capi, synth
from paramugsy_amd.translate import Workload


def case_paths(name):
    case = os.path.join(GOLDEN, "translate_" + name)
    with open(os.path.join(case, "nucmer.list")) as f:
        deltas = [os.path.join(case, ln.strip()) for ln in f if ln.strip()]
    return os.path.join(case, "profiles-l"), os.path.join(case, "profiles-r"), deltas


@pytest.mark.parametrize("name", ["typical", "gappy", "reverse", "tiny_blocks", "empty"])
def test_unit_list_matches_oracle(name, oracle_build):
    import pyoracle
    l, r, deltas = case_paths(name)
    t = Workload.load(l, r, deltas).tables()
    ou = pyoracle.enumerate_units(l, r, deltas)
    for k in ("delta", "left", "right"):
        assert np.array_equal(ou[k], t.units[k]), k
    if name != "empty":
        assert t.n_units > 20


def test_rows_table_matches_profiles_file():
    l, r, deltas = case_paths("gappy")
    t = Workload.load(l, r, deltas).tables()
    # re-read the profiles file independently
    starts, ends, lens, ngaps = [], [], [], []
    with open(os.path.join(l, "profiles")) as f:
        lines = f.read().split("\n")
    i = 0
    while i < len(lines) and lines[i]:
        head = lines[i].split(" ")
        starts.append(int(head[3])); ends.append(int(head[4])); lens.append(int(head[5]))
        i += 1
        n = 0
        while lines[i] != "0":
            n += 1
            i += 1
        ngaps.append(n)
        i += 2
    assert t.left["start"].tolist() == starts
    assert t.left["end"].tolist() == ends
    assert t.left["length"].tolist() == lens
    assert np.diff(t.left["gap_off"]).tolist() == ngaps


def test_delta_known_answer_through_product_parser(tmp_path):
    # lib/profiles_lib/m_delta.cc:43-49
    d = tmp_path / "ka.delta"
    d.write_text("/a /b\nNUCMER\n>r q 5000 5000\n1 2000 1 2000 0 0 0\n106\n-6\n1797\n-9\n-9\n-1\n7\n1\n0\n")
    (tmp_path / "l").mkdir()
    (tmp_path / "r").mkdir()
    (tmp_path / "l" / "profiles").write_text("")
    (tmp_path / "r" / "profiles").write_text("")
    t = Workload.load(str(tmp_path / "l"), str(tmp_path / "r"), [str(d)]).tables()
    assert list(zip(t.deltas["ref_gap_start"], t.deltas["ref_gap_end"])) == [(112, 112), (1918, 1918), (1927, 1928)]
    assert list(zip(t.deltas["qry_gap_start"], t.deltas["qry_gap_end"])) == [(106, 106), (1909, 1909), (1935, 1936)]


def test_malformed_delta_is_a_parse_error(tmp_path):
    (tmp_path / "l").mkdir()
    (tmp_path / "r").mkdir()
    (tmp_path / "l" / "profiles").write_text("")
    (tmp_path / "r" / "profiles").write_text("")
    bad = tmp_path / "bad.delta"
    bad.write_text("/a /b\nNUCMER\n>r q 10 10\n1 2 3\n")
    with pytest.raises(capi.PmError) as e:
        Workload.load(str(tmp_path / "l"), str(tmp_path / "r"), [str(bad)])
    assert e.value.code == capi.PM_E_PARSE
    with pytest.raises(capi.PmError) as e:
        Workload.load(str(tmp_path / "l"), str(tmp_path / "r"), [str(tmp_path / "missing.delta")])
    assert e.value.code == capi.PM_E_PARSE


def test_malformed_profiles_is_a_parse_error(tmp_path):
    (tmp_path / "l").mkdir()
    (tmp_path / "r").mkdir()
    (tmp_path / "l" / "profiles").write_text("l.l_0000 0 g 1 10 4294967296 100\n0\nAAAAAAAAAA\n")  # p_length > unsigned int
    (tmp_path / "r" / "profiles").write_text("")
    with pytest.raises(capi.PmError) as e:
        Workload.load(str(tmp_path / "l"), str(tmp_path / "r"), [])
    assert e.value.code == capi.PM_E_PARSE
