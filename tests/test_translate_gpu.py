"""Parity of the HIP translate path with the oracle and with the golden fixtures the upstream reference
produced.  Everything here calls through the C ABI of libparamugsy_amd.so.  Bit-exact: integer coordinates."""
import filecmp
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from paramugsy_amd import capi, synth
from paramugsy_amd.translate import TranslateJob, Workload, translate, profile_idx_of_seq_idx, seq_idx_of_profile_idx

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["int", "int64"])
def coordinate_width(request, monkeypatch):
    """Every test of this file runs twice: on the int tables a job takes when all its numbers are below 2^25 (all the
    inputs here are), and with PM_TRANSLATE_WIDE=1, on the int64 tables."""  # (and see below: the prefix sums)
    monkeypatch.setenv("PM_TRANSLATE_WIDE", "1" if request.param == "int64" else "0")
    # the int64 runs also take the library's prefix sums (what jobs above 8.4 M units use), the int runs the step's own
    monkeypatch.setenv("PM_TRANSLATE_LIBRARY_SCANS", "1" if request.param == "int64" else "0")
    return request.param


CASES = ["typical", "gappy", "reverse", "tiny_blocks", "empty"]


def case_paths(name):
    case = os.path.join(GOLDEN, "translate_" + name)
    with open(os.path.join(case, "nucmer.list")) as f:
        deltas = [ln.strip() for ln in f if ln.strip()]
    return case, deltas


@pytest.mark.parametrize("name", CASES)
def test_file_level_bytes_equal_reference_golden(name, tmp_path):
    case, deltas = case_paths(name)
    out = str(tmp_path / "out.delta")
    cwd = os.getcwd()
    os.chdir(case)  # the first output line echoes the directory arguments
    try:
        translate("profiles-l", "profiles-r", deltas, out)
    finally:
        os.chdir(cwd)
    assert filecmp.cmp(out, os.path.join(case, "expected.delta"), shallow=False)


@pytest.mark.parametrize("name", ["typical", "reverse"])
def test_cli_drop_in_bytes_equal_reference_golden(name, tmp_path):
    case, _ = case_paths(name)
    exe = os.path.join(ROOT, "bin", "m_translate")
    out = str(tmp_path / "cli.delta")
    r = subprocess.run([exe, "profiles-l", "profiles-r", "nucmer.list", out], cwd=case, capture_output=True)
    assert r.returncode == 0, r.stderr
    assert filecmp.cmp(out, os.path.join(case, "expected.delta"), shallow=False)
    r = subprocess.run([exe, "profiles-l"], capture_output=True)  # m_translate_main.cc:22-25
    assert r.returncode == 1 and b"Usage: m_translate" in r.stderr


def assert_same_result(res, ora):
    assert np.array_equal(res.status, ora["status"])
    assert np.array_equal(res.unit_entry_off, ora["unit_entry_off"])
    assert len(res.entries) == len(ora["entries"])
    for k in ("ref_start", "ref_end", "qry_start", "qry_end", "n_offsets"):
        assert np.array_equal(res.entries[k], ora["entries"][k]), k
    # offsets: compare per entry through each side's own offset_begin (layouts agree, but do not rely on it)
    assert np.array_equal(res.entries["offset_begin"], ora["entries"]["offset_begin"])
    assert np.array_equal(res.offsets, ora["offsets"])


MODES = {
    "typical": dict(),
    "gappy": dict(gap_rate=0.05, mean_gap=6.0, indel_rate=0.02, mean_indel=4.0, adjacent_prob=0.1, edge_gap_prob=0.5),
    "reverse": dict(genome_len=8000, n_blocks=30, mean_cols=120, gap_rate=0.08, indel_rate=0.05, mean_len=400,
                    entries_per_delta=80, rev_prob=0.5, delta_rev_prob=0.5, spacing=5),
    "tiny_blocks": dict(genome_len=3000, n_blocks=150, mean_cols=8, gap_rate=0.1, mean_gap=3.0, indel_rate=0.05, mean_indel=8.0,
                        mean_len=150, entries_per_delta=60, spacing=3, edge_gap_prob=0.4, adjacent_prob=0.15, delta_rev_prob=0.4),
    "long_rows": dict(n_left=2, n_right=2, genome_len=400000, n_blocks=12, mean_cols=20000, gap_rate=0.02, mean_len=6000,
                      entries_per_delta=60, spacing=200),
}


@pytest.mark.parametrize("mode", sorted(MODES))
@pytest.mark.parametrize("seed", [21, 22])
def test_job_level_equals_oracle(mode, seed, oracle_build, tmp_path):
    import pyoracle
    w = synth.make_workload(str(tmp_path / "job"), seed * 100 + len(mode), **MODES[mode])
    t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    assert t.n_units > 10
    job = TranslateJob(t)
    job.run()
    res = job.fetch()
    ora = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
    assert_same_result(res, ora)
    assert res.all_ok and len(res.entries) > 0
    # idempotence: a second pass over the resident batch gives the same bytes
    job.run()
    res2 = job.fetch()
    assert np.array_equal(res2.offsets, res.offsets) and np.array_equal(res2.entries, res.entries)
    job.close()


def corrupt_tables(t, rng):
    """Keeps every gap list ascending and disjoint (inside the library's domain) but makes the tables
    inconsistent: rows with more gap columns than p_length allows, shifted gaps, short p_length, stretched
    delta gaps.  The reference dies in some of the resulting units."""
    for rows in (t.left, t.right):
        off = rows["gap_off"]
        for r in np.nonzero(rng.random(len(off) - 1) < 0.25)[0]:
            a, b = int(off[r]), int(off[r + 1])
            if b <= a:
                continue
            k, sh, kind = int(rng.integers(0, b - a)), int(rng.integers(1, 25)), int(rng.integers(0, 3))
            if kind == 0:
                rows["gap_end"][a + k:b] += sh
                rows["gap_start"][a + k + 1:b] += sh
            elif kind == 1:
                rows["gap_start"][a:b] += sh
                rows["gap_end"][a:b] += sh
            else:
                rows["length"][r] = max(1, rows["length"][r] - sh)
    # stretch one gap of an entry and move every later gap of BOTH rows along with it: the entry's columns stay
    # a valid alignment (no column gapped in both rows), only its ranges no longer fit them
    D = t.deltas
    for d in np.nonzero(rng.random(len(D["ref_start"])) < 0.2)[0]:
        name = ("ref", "qry")[int(rng.integers(0, 2))]
        a, b = int(D[name + "_gap_off"][d]), int(D[name + "_gap_off"][d + 1])
        if b <= a:
            continue
        k, sh = a + int(rng.integers(0, b - a)), int(rng.integers(1, 10))
        pivot = D[name + "_gap_end"][k]
        D[name + "_gap_end"][k] += sh
        for other in ("ref", "qry"):
            oa, ob = int(D[other + "_gap_off"][d]), int(D[other + "_gap_off"][d + 1])
            later = np.arange(oa, ob)[D[other + "_gap_start"][oa:ob] > pivot]
            D[other + "_gap_start"][later] += sh
            D[other + "_gap_end"][later] += sh


@pytest.mark.parametrize("seed", [4242, 4243, 4244])
def test_failure_classes_equal_oracle(seed, oracle_build, tmp_path):
    """Units the reference would die in (exceptions / asserts): same class per unit, and the entries emitted
    before the failure are the same."""
    import pyoracle
    w = synth.make_workload(str(tmp_path / "job"), seed, **MODES["reverse"])
    t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    corrupt_tables(t, np.random.default_rng(seed))
    job = TranslateJob(t)
    job.run()
    res = job.fetch()
    ora = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
    assert_same_result(res, ora)
    assert not res.all_ok and (res.status != 0).sum() >= 3


def test_column_gapped_in_both_rows_prints_what_the_reference_prints(oracle_build):
    """A delta entry with the same column gapped in its reference row AND its query row cannot come out of a
    delta file (m_delta.cc:50-68 hands out columns once) and makes the reference's writer emit a stray 0 (its two-list merge
    takes the query gap first and then finds the reference gap at distance 0).  The builder's gaps arrive out of the writer's
    merge order here; the FIX pass merges them as the writer does, stray 0 included."""
    import pyoracle
    from paramugsy_amd.translate import Tables
    i64 = lambda *v: np.array(v, dtype=np.int64)
    left = {"start": i64(1), "end": i64(100), "length": i64(100), "gap_off": i64(0, 0), "gap_start": i64(), "gap_end": i64()}
    right = {"start": i64(1), "end": i64(100), "length": i64(100), "gap_off": i64(0, 0), "gap_start": i64(), "gap_end": i64()}
    deltas = {"ref_start": i64(1), "ref_end": i64(99), "qry_start": i64(1), "qry_end": i64(99),
              "ref_gap_off": i64(0, 1), "ref_gap_start": i64(50), "ref_gap_end": i64(50),
              "qry_gap_off": i64(0, 1), "qry_gap_start": i64(50), "qry_gap_end": i64(50)}
    z = np.zeros(1, dtype=np.int32)
    t = Tables(left, right, deltas, {"delta": z, "left": z, "right": z})
    job = TranslateJob(t)
    job.run()
    res = job.fetch()
    assert_same_result(res, pyoracle.translate_units(t.left, t.right, t.deltas, t.units))
    assert res.status.tolist() == [0] and 0 in res.offsets[:-1].tolist()
    job.close()


@pytest.mark.parametrize("seed,mode", [(33, "typical"), (188, "reverse"), (208, "typical"), (219, "gappy"), (298, "reverse")])
def test_gaps_out_of_the_writers_order_are_merged_as_the_writer_does(seed, mode, oracle_build, tmp_path):
    """Inconsistent tables whose units hand the builder gaps in another order than the writer's merge takes them (seeds found by
    tools/translate_fuzz.py; the library used to refuse such units): equal to the oracle, unit by unit, on both table widths."""
    import pyoracle
    rng = np.random.default_rng(seed)
    kw = dict(MODES[mode])
    assert sorted(MODES)[int(rng.integers(0, len(MODES)))] == mode  # the same draws as the fuzz tool makes for this seed
    for key, lo, hi in (("gap_rate", 0.0, 0.15), ("indel_rate", 0.0, 0.08), ("rev_prob", 0.0, 0.6), ("delta_rev_prob", 0.0, 0.6),
                        ("edge_gap_prob", 0.0, 0.6), ("adjacent_prob", 0.0, 0.2)):
        if rng.random() < 0.5:
            kw[key] = float(rng.uniform(lo, hi))
    if mode != "long_rows" and rng.random() < 0.5:
        kw["entries_per_delta"] = int(rng.integers(5, 200))
    w = synth.make_workload(str(tmp_path / "job"), seed, **kw)
    t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    if rng.random() < 0.5:
        corrupt_tables(t, rng)
    ora = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
    for wide in ("0", "1"):
        os.environ["PM_TRANSLATE_WIDE"] = wide
        try:
            job = TranslateJob(t)
        finally:
            del os.environ["PM_TRANSLATE_WIDE"]
        job.run()
        res = job.fetch()
        assert_same_result(res, ora)
        job.run()  # and again: the FIX pass rewrites what the EMIT pass wrote, every pass
        assert_same_result(job.fetch(), ora)
        job.close()


def test_prefix_sums_of_the_step_equal_the_librarys_over_many_tiles(tmp_path, monkeypatch):
    """The step's own prefix sums (tiles of 2 048 units: the live list, the entry and offset places) against the library scans a job
    above 8.4 M units keeps, on a job of some hundred tiles whose last tile is ragged: the same live order, places and outputs."""
    w = synth.make_workload(str(tmp_path / "job"), 77, n_left=3, n_right=3, genome_len=200000, n_blocks=500, n_deltas=4,
                            entries_per_delta=1500, mean_len=1500)
    t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    assert t.n_units > 40 * 2048 and t.n_units % 2048 != 0
    got = []
    for library in ("0", "1"):
        monkeypatch.setenv("PM_TRANSLATE_LIBRARY_SCANS", library)
        job = TranslateJob(t)
        job.run()
        job.run()
        got.append(job.fetch())
        job.close()
    a, b = got
    assert np.array_equal(a.status, b.status) and np.array_equal(a.unit_entry_off, b.unit_entry_off)
    assert a.entries.tobytes() == b.entries.tobytes() and np.array_equal(a.offsets, b.offsets)
    assert len(a.offsets) > 0


@pytest.mark.parametrize("mode,overlap", [(m, 0.0) for m in sorted(MODES)] + [("typical", 0.3), ("tiny_blocks", 0.8), ("reverse", 0.05)])
def test_unit_list_made_on_the_device_equals_the_hosts_loops(mode, overlap, oracle_build, tmp_path):
    """pm_job_create_from_workload lists the units on the device (what every file-level entry runs on): the same triples in the same
    order as the host's restatement of the loops at m_translate.cc:666-707, also when a genome's rows overlap and nest (their ends are
    then out of order under the index's sort by start, and the binary search has to land where libstdc++'s lands) -- and the job run
    over them equals the oracle."""
    import pyoracle
    kw = dict(MODES[mode])
    if overlap:
        kw["overlap_prob"] = overlap
    w = synth.make_workload(str(tmp_path / "job"), 5150 + len(mode), **kw)
    wl = Workload.load(w.left_dir, w.right_dir, w.delta_paths)
    t = wl.tables()
    job = TranslateJob.from_workload(wl)
    assert job.n_units == t.n_units > 10
    got = job.units()
    for k in ("delta", "left", "right"):
        assert np.array_equal(got[k], t.units[k]), k
    if overlap:  # some entry does see several rows of one genome
        assert (np.bincount(t.units["delta"]) > 1).any()
    job.run()
    assert_same_result(job.fetch(), pyoracle.translate_units(t.left, t.right, t.deltas, t.units))
    job.close()


@pytest.mark.parametrize("name", ["typical", "gappy", "reverse", "tiny_blocks", "empty"])
def test_unit_list_made_on_the_device_on_the_golden_jobs(name):
    """The five golden jobs (the last one has no work units at all): the device's list = the host's, and a job of zero units runs."""
    case = os.path.join(GOLDEN, "translate_" + name)
    with open(os.path.join(case, "nucmer.list")) as f:
        deltas = [os.path.join(case, ln.strip()) for ln in f if ln.strip()]
    wl = Workload.load(os.path.join(case, "profiles-l"), os.path.join(case, "profiles-r"), deltas)
    t = wl.tables()
    job = TranslateJob.from_workload(wl)
    assert job.n_units == t.n_units
    got = job.units()
    for k in ("delta", "left", "right"):
        assert np.array_equal(got[k], t.units[k]), k
    job.run()
    res = job.fetch()
    assert len(res.status) == t.n_units and res.all_ok
    job.close()


def test_unit_list_on_the_device_with_sequences_a_side_does_not_have(tmp_path):
    """Entries whose reference or query sequence has no rows on its side yield no units (m_translate.cc:676-681)."""
    w = synth.make_workload(str(tmp_path / "job"), 99, n_left=3, n_right=3, row_prob=1.0)
    # a delta file naming genomes the sides do not hold, between two that they do
    text = open(w.delta_paths[0]).read()
    stranger = text.replace(">L0.chr ", ">nobody.chr ").replace(" R1.chr ", " nothing.chr ")
    extra = str(tmp_path / "job" / "strangers.delta")
    open(extra, "w").write(stranger)
    paths = [w.delta_paths[0], extra] + w.delta_paths[1:]
    wl = Workload.load(w.left_dir, w.right_dir, paths)
    t = wl.tables()
    job = TranslateJob.from_workload(wl)
    got = job.units()
    assert job.n_units == t.n_units > 0
    for k in ("delta", "left", "right"):
        assert np.array_equal(got[k], t.units[k]), k
    job.close()


def test_malformed_gap_lists_are_refused_not_miscomputed(tmp_path):
    w = synth.make_workload(str(tmp_path / "job"), 77)
    t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    # swap two gaps of the first left row that has at least two: no longer ascending
    off = t.left["gap_off"]
    rows = [int(r) for r in np.unique(t.units["left"]) if off[r + 1] - off[r] >= 2]
    r = rows[0]
    a = int(off[r])
    for k in ("gap_start", "gap_end"):
        t.left[k][[a, a + 1]] = t.left[k][[a + 1, a]]
    job = TranslateJob(t)
    job.run()
    res = job.fetch()
    touched = t.units["left"] == r
    assert touched.any()
    assert (res.status[touched] == capi.PM_ST_MALFORMED_INPUT).all()
    assert (res.status[~touched] == 0).all()


def test_empty_batches():
    z = np.zeros(0, dtype=np.int64)
    z1 = np.zeros(1, dtype=np.int64)
    rows = {"start": z, "end": z, "length": z, "gap_off": z1, "gap_start": z, "gap_end": z}
    deltas = {k: z for k in ("ref_start", "ref_end", "qry_start", "qry_end", "ref_gap_start", "ref_gap_end", "qry_gap_start", "qry_gap_end")}
    deltas["ref_gap_off"] = z1
    deltas["qry_gap_off"] = z1
    from paramugsy_amd.translate import Tables
    zi = np.zeros(0, dtype=np.int32)
    job = TranslateJob(Tables(rows, rows, deltas, {"delta": zi, "left": zi, "right": zi}))
    job.run()
    res = job.fetch()
    assert len(res.entries) == 0 and len(res.offsets) == 0 and res.unit_entry_off.tolist() == [0]


def test_batched_index_conversions_equal_oracle(oracle_build, tmp_path):
    import pyoracle
    w = synth.make_workload(str(tmp_path / "job"), 31, **MODES["gappy"])
    t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    rows = t.left
    rng = np.random.default_rng(5)
    n = 20000
    row = rng.integers(0, len(rows["start"]), size=n).astype(np.int32)
    lo = np.minimum(rows["start"], rows["end"])[row]
    hi = np.maximum(rows["start"], rows["end"])[row]
    si = rng.integers(lo - 3, hi + 4)
    pi = rng.integers(-2, rows["length"][row] + 4)
    a, sa = profile_idx_of_seq_idx(rows, row, si)
    b, sb = pyoracle.profile_idx_of_seq_idx(rows, row, si)
    assert np.array_equal(sa, sb) and np.array_equal(a[sa == 0], b[sb == 0])
    assert (sa != 0).any() and (sa == 0).any()
    a, sa = seq_idx_of_profile_idx(rows, row, pi)
    b, sb = pyoracle.seq_idx_of_profile_idx(rows, row, pi)
    assert np.array_equal(sa, sb) and np.array_equal(a[sa == 0], b[sb == 0])
    assert set(sa.tolist()) >= {0, capi.PM_ST_IS_NONE, capi.PM_ST_PROFILE_IDX_OUT_OF_RANGE}
    # round trip, a size-independent property: every base maps to a column that maps back to it
    ok = (sa == 0) & (pi >= 1)  # a4 does not reject pi <= 0 (m_profile.cc:115); those map outside the row
    c, sc = profile_idx_of_seq_idx(rows, row[ok], a[ok])
    assert (sc == 0).all() and np.array_equal(c, pi[ok])


def test_full_size_job_bytes_equal_cpu_side(oracle_build, tmp_path):
    """The bench workload at full size (1.45 M work units, 64 MB of delta text) through the drop-in executable,
    byte for byte against the upstream binary when it travelled with the snapshot (oracle/_ref), else the oracle."""
    w = synth.make_workload(str(tmp_path / "job"), 20261003, n_left=4, n_right=4, genome_len=1000000, n_blocks=2500, n_deltas=16,
                            entries_per_delta=6000, mean_len=1500)
    ref = os.path.join(ROOT, "oracle", "_ref", "m_translate")
    cpu = ref if os.path.exists(ref) else os.path.join(oracle_build, "oracle_m_translate")
    a, b = str(tmp_path / "cpu.delta"), str(tmp_path / "gpu.delta")
    assert subprocess.run([cpu, w.left_dir, w.right_dir, w.list_path, a]).returncode == 0
    assert subprocess.run([os.path.join(ROOT, "bin", "m_translate"), w.left_dir, w.right_dir, w.list_path, b]).returncode == 0
    assert os.path.getsize(a) > 50_000_000
    assert filecmp.cmp(a, b, shallow=False)


def test_coordinate_width_is_chosen_from_the_tables(coordinate_width, oracle_build, tmp_path):
    """Numbers below 2^25 -> int kernels (unless PM_TRANSLATE_WIDE=1).  Sequence POSITIONS at 2^25 or above, with every row, entry and
    gap list still short -> the int kernels with 64-bit positions (round 5: a position enters the arithmetic only as its distance
    from the start of the row that contains it).  A length, span or gap column at 2^25 or above -> int64 kernels.  Either way the
    oracle's answer -- and, translate's output being columns, the same answer wherever along its sequences the job lies."""
    import pyoracle
    w = synth.make_workload(str(tmp_path / "job"), 5, **MODES["typical"])
    t0 = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    first = None
    for lshift, rshift, long_row, bits_if_free, pos_bits_if_free in (
            ((1 << 25) - 70000, (1 << 25) - 70000, False, 32, 32), ((1 << 25) + 5, (1 << 25) + 5, False, 32, 64),
            (1 << 40, 0, False, 32, 64), (0, (1 << 33) + 9, False, 32, 64), ((1 << 61) + 3, 1 << 47, False, 32, 64),
            (1 << 62, 0, False, 64, 64),  # differences of such positions could wrap: nothing is assumed about them
            (1 << 40, 1 << 40, True, 64, 64)):
        t = synth.shift_positions(t0, lshift, rshift)
        if long_row:  # one row (no unit uses it) that spans 2^25 bases: the columns no longer fit, the job is the int64 job
            for k, v in (("start", 1), ("end", (1 << 25) + 10), ("length", (1 << 25) + 10)):
                t.left[k] = np.append(t.left[k], np.int64(v))
            t.left["gap_off"] = np.append(t.left["gap_off"], t.left["gap_off"][-1])
        job = TranslateJob(t)
        try:
            forced = coordinate_width == "int64"
            assert job.coordinate_bits() == (64 if forced else bits_if_free)
            assert job.position_bits() == (64 if forced else pos_bits_if_free)
            job.run()
            res = job.fetch()
            job.run()  # (and again: the emit pass of the job with wide positions is the int job's, from the saved states)
            res2 = job.fetch()
        finally:
            job.close()
        ora = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
        assert_same_result(res, ora)
        assert_same_result(res2, ora)
        assert (res.status == 0).all() and len(res.entries) > 10
        if first is None:
            first = res
        assert res.entries.tobytes() == first.entries.tobytes() and np.array_equal(res.offsets, first.offsets)


def test_wide_positions_through_the_files_and_the_failure_classes(coordinate_width, oracle_build, tmp_path):
    """The job with 64-bit positions (int columns) on tables that contradict themselves: the same failure classes as the oracle, unit by
    unit, as for the other two widths (test_failure_classes_equal_oracle); and on every mode of the generator."""
    import pyoracle
    for seed, mode in ((4242, "reverse"), (4243, "tiny_blocks"), (77, "gappy")):
        w = synth.make_workload(str(tmp_path / ("job%d" % seed)), seed, **MODES[mode])
        t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
        if seed != 77:
            corrupt_tables(t, np.random.default_rng(seed))
        t = synth.shift_positions(t, (1 << 36) + 11, (1 << 52) + 5)
        job = TranslateJob(t)
        try:
            assert job.position_bits() == 64 and job.coordinate_bits() == (64 if coordinate_width == "int64" else 32)
            job.run()
            res = job.fetch()
        finally:
            job.close()
        ora = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
        assert_same_result(res, ora)
        assert (res.status != 0).any() == (seed != 77)


def test_file_level_failure_writes_what_precedes_it(tmp_path):
    """A job in which some unit fails (here: a row whose recorded profile length is too short for its gaps): the
    reference dies inside that unit; the drop-in writes the output that precedes it -- byte for byte the beginning of
    what the intact job prints, up to an entry boundary -- then reports the unit and exits 134."""
    w = synth.make_workload(str(tmp_path / "job"), 31, **MODES["typical"])
    good = str(tmp_path / "good.delta")
    translate(w.left_dir, w.right_dir, w.delta_paths, good)
    full = open(good, "rb").read()
    assert full.count(b"\n") > 200
    # break one row in the later part of the left side: halve its p_length field
    path = os.path.join(w.left_dir, "profiles")
    lines = open(path).read().split("\n")
    heads = [i for i, ln in enumerate(lines) if ln.count(" ") == 6 and not ln[0].isdigit()]
    broke = 0
    for i in heads[len(heads) * 2 // 3:]:
        f = lines[i].split(" ")
        f[5] = str(max(1, int(f[5]) // 2))
        lines[i] = " ".join(f)
        broke += 1
    open(path, "w").write("\n".join(lines))
    assert broke > 3
    bad = str(tmp_path / "bad.delta")
    with pytest.raises(capi.PmError) as e:
        translate(w.left_dir, w.right_dir, w.delta_paths, bad)
    assert e.value.code == capi.PM_E_UNIT and "work unit" in str(e.value)
    part = open(bad, "rb").read()
    assert 100 < len(part) < len(full) and full.startswith(part)
    assert part.endswith(b"\n0\n") or part.count(b"\n") == 2  # ends on an entry boundary
    # the executable: same bytes, exit status 134 (the reference ends in SIGABRT)
    cli = str(tmp_path / "cli.delta")
    r = subprocess.run([os.path.join(ROOT, "bin", "m_translate"), w.left_dir, w.right_dir, w.list_path, cli], capture_output=True)
    assert r.returncode == 134 and b"work unit" in r.stderr
    assert open(cli, "rb").read() == part


def test_explicit_options_per_job_on_every_thread_and_device(tmp_path, monkeypatch):
    """pm_translate_options_t (round 5): a job takes its options when it is created or run -- also over a device list, whose workers
    are threads that never look at the environment or at the defaults of another job -- and two jobs created side by side under
    different options each keep their own."""
    import ctypes as C
    import threading
    for name in ("PM_TRANSLATE_WIDE", "PM_TRANSLATE_LIBRARY_SCANS", "PM_NO_SOA", "PM_TIMING"):
        monkeypatch.delenv(name, raising=False)
    case, deltas = case_paths("typical")
    l = capi.lib()
    arr = (C.c_char_p * len(deltas))(*[os.path.join(case, p).encode() for p in deltas])
    devs = (C.c_int32 * 3)(0, 0, 0)
    expected = open(os.path.join(case, "expected.delta"), "rb").read()
    for bits, scans, no_soa in ((0, 0, 0), (64, 1, 1), (32, 0, 1)):
        opt = capi.PmTranslateOptions()
        opt.coordinate_bits, opt.library_scans, opt.no_side_file = bits, scans, no_soa
        for n_dev in (1, 3):
            out = str(tmp_path / ("o_%d_%d.delta" % (bits, n_dev)))
            capi.check(l.pm_translate_files_opt(os.path.join(case, "profiles-l").encode(), os.path.join(case, "profiles-r").encode(), arr,
                                                len(deltas), out.encode(), b"profiles-l", b"profiles-r", devs, n_dev, C.byref(opt)))
            assert open(out, "rb").read() == expected, (bits, n_dev)
    bad = capi.PmTranslateOptions()
    bad.coordinate_bits = 16
    assert l.pm_translate_files_opt(b"a", b"b", arr, 0, str(tmp_path / "x").encode(), b"a", b"b", devs, 1, C.byref(bad)) == capi.PM_E_INVALID
    # two jobs at once, each with its own width
    t = Workload.load(os.path.join(case, "profiles-l"), os.path.join(case, "profiles-r"), [os.path.join(case, p) for p in deltas]).tables()
    got = {}

    def make(bits):
        opt = capi.PmTranslateOptions()
        opt.coordinate_bits = bits
        for _ in range(5):
            job = TranslateJob(t, options=opt)
            got.setdefault(bits, set()).add(job.coordinate_bits())
            job.close()
    ts = [threading.Thread(target=make, args=(b,)) for b in (0, 64)]
    for th in ts:
        th.start()
    for th in ts:
        th.join()
    assert got == {0: {32}, 64: {64}}
    # the process's defaults are what the entries without an options argument take
    wide = capi.PmTranslateOptions()
    wide.coordinate_bits = 64
    capi.check(l.pm_translate_set_default_options(C.byref(wide)))
    try:
        h = C.c_void_p()
        ls, k1 = capi.rows_struct(t.left)
        rs, k2 = capi.rows_struct(t.right)
        ds, k3 = capi.deltas_struct(t.deltas)
        us, k4 = capi.units_struct(t.units)
        capi.check(l.pm_job_create(C.byref(ls), C.byref(rs), C.byref(ds), C.byref(us), 0, C.byref(h)))
        b = C.c_int()
        capi.check(l.pm_job_coordinate_bits(h, C.byref(b)))
        l.pm_job_destroy(h)
        assert b.value == 64
    finally:
        capi.check(l.pm_translate_set_default_options(None))
