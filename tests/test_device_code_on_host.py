"""The device-side unit arithmetic (paramugsy_amd/csrc/translate_device.hpp, __host__ __device__) compiled for
the host by tests/tools/unit_host_harness.cpp and compared with the oracle.  CPU only.  This is a test of the
kernel's code, not a product path: the library never calls these functions on the host.

Set PM_HARNESS_SANITIZE=1 to build the harness with -fsanitize=address,undefined (then run pytest with
LD_PRELOAD=<libclang_rt.asan-x86_64.so> ASAN_OPTIONS=detect_leaks=0)."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from paramugsy_amd import capi, synth
from paramugsy_amd.translate import Workload


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    if shutil.which("hipcc") is None:
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("harness") / "libunit_host.so")
    cmd = ["hipcc", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-Wno-option-ignored", "-o", out,
           os.path.join(ROOT, "tests", "tools", "unit_host_harness.cpp")]
    if os.environ.get("PM_HARNESS_SANITIZE") == "1":
        cmd[1:1] = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    subprocess.run(cmd, check=True, capture_output=True)
    h = C.CDLL(out)
    for fn in (h.unit_host_run, h.unit_host_run_narrow, h.unit_host_run_wide_positions):
        fn.argtypes = [C.c_void_p] * 4 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.c_int64,
                       C.c_void_p, C.c_int64]
    return h


PM_ST_NARROW = 100  # internal status of the int instantiation: "redo this job in int64"


WIDTHS = [False, True, "positions"]  # int64; int; int columns with 64-bit sequence positions (round 5)
LIMIT = 1 << 25  # PM_NARROW_INPUT_LIMIT


def columns_fit_int(t):
    """What pm_job_create checks before it takes the wide-positions job: every length, span and gap column below the limit."""
    big = 0
    for side in (t.left, t.right):
        for k in ("length", "gap_start", "gap_end"):
            big = max(big, int(np.abs(side[k]).max(initial=0)))
        big = max(big, int((np.abs(side["end"] - side["start"]) + 1).max(initial=0)), int(side["length"].sum()))
    for a, b in (("ref_start", "ref_end"), ("qry_start", "qry_end")):
        big = max(big, int((np.abs(t.deltas[b] - t.deltas[a]) + 1).max(initial=0)))
    for k in ("ref_gap_start", "ref_gap_end", "qry_gap_start", "qry_gap_end"):
        big = max(big, int(np.abs(t.deltas[k]).max(initial=0)))
    return big < LIMIT // 4  # (with room for the prefix sums, which the library checks one by one)


def host_run(h, t, narrow=False):
    run = {False: h.unit_host_run, True: h.unit_host_run_narrow, "positions": h.unit_host_run_wide_positions}[narrow]
    ls, k1 = capi.rows_struct(t.left)
    rs, k2 = capi.rows_struct(t.right)
    ds, k3 = capi.deltas_struct(t.deltas)
    us, k4 = capi.units_struct(t.units)
    U = t.n_units
    st = np.zeros(U, np.int32)
    eo = np.zeros(U + 1, np.int64)
    ne, no = C.c_int64(), C.c_int64()
    ent = np.zeros(1, capi.ENTRY_DTYPE)
    off = np.zeros(1, np.int64)
    args = (C.byref(ls), C.byref(rs), C.byref(ds), C.byref(us), st.ctypes.data, eo.ctypes.data, C.byref(ne), C.byref(no))
    run(*args, ent.ctypes.data, 0, off.ctypes.data, 0)
    ent = np.zeros(max(1, ne.value), capi.ENTRY_DTYPE)
    off = np.zeros(max(1, no.value), np.int64)
    rc = run(*args, ent.ctypes.data, ne.value, off.ctypes.data, no.value)
    assert rc == 0, "count and emit passes disagree" if rc == 2 else rc
    return st, eo, ent[:ne.value], off[:no.value]


def check_against_oracle(h, t, narrow=False):
    """narrow: the int instantiation of the same code (the fast path for tables below 2^25).  It must either give the
    oracle's answer on every unit, or report PM_ST_NARROW somewhere (the library then redoes the job in int64)."""
    import pyoracle
    ora0 = None
    if narrow == "positions":
        # the same job 2^40 bases along its left sequences and 2^33 along its right ones: the positions need the `long`, the columns do
        # not -- and what the job writes (columns) is what the unmoved job writes
        if not columns_fit_int(t):
            return None  # the library would not take this instantiation for such tables
        ora0 = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
        t = synth.shift_positions(t, (1 << 40) + 12345, (1 << 33) + 777)
    st, eo, ent, off = host_run(h, t, narrow)
    if narrow and (st == PM_ST_NARROW).any():
        return None
    ora = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
    if ora0 is not None:
        assert np.array_equal(ora0["status"], ora["status"]) and np.array_equal(ora0["offsets"], ora["offsets"])
        assert ora0["entries"].tobytes() == ora["entries"].tobytes()
    assert np.array_equal(st, ora["status"])
    assert np.array_equal(eo, ora["unit_entry_off"])
    assert np.array_equal(off, ora["offsets"])
    for k in ("ref_start", "ref_end", "qry_start", "qry_end", "offset_begin", "n_offsets"):
        assert np.array_equal(ent[k], ora["entries"][k]), k
    return st


@pytest.mark.parametrize("narrow", WIDTHS)
@pytest.mark.parametrize("name", ["typical", "gappy", "reverse", "tiny_blocks", "empty"])
def test_device_code_equals_oracle_on_golden_inputs(name, narrow, harness, oracle_build):
    case = os.path.join(GOLDEN, "translate_" + name)
    with open(os.path.join(case, "nucmer.list")) as f:
        deltas = [os.path.join(case, ln.strip()) for ln in f if ln.strip()]
    t = Workload.load(os.path.join(case, "profiles-l"), os.path.join(case, "profiles-r"), deltas).tables()
    st = check_against_oracle(harness, t, narrow)
    assert st is not None and (st == 0).all()  # sane tables never trip the int path's range check


@pytest.mark.parametrize("narrow", WIDTHS)
@pytest.mark.parametrize("seed", range(4242, 4250))
def test_device_code_equals_oracle_on_inconsistent_tables(seed, narrow, harness, oracle_build, tmp_path):
    from test_translate_gpu import MODES, corrupt_tables
    w = synth.make_workload(str(tmp_path / "job"), seed, **MODES["reverse" if seed % 2 == 0 else "tiny_blocks"])
    t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    corrupt_tables(t, np.random.default_rng(seed))
    st = check_against_oracle(harness, t, narrow)
    assert st is None or (st != 0).any()


@pytest.mark.parametrize("narrow", WIDTHS)
@pytest.mark.parametrize("seed,mode", [(33, "typical"), (188, "reverse"), (208, "typical"), (219, "gappy"), (298, "reverse")])
def test_gaps_out_of_the_writers_order_are_merged_as_the_writer_does(seed, mode, narrow, harness, oracle_build, tmp_path):
    """Tables that contradict themselves can hand the builder gaps in another order than the writer's two-list merge
    (m_delta_stream_writer.hh:14-53) takes them; the unit's offsets are then not what on-the-fly emission gives.  The FIX pass
    records such a unit's gaps and merges them as the writer does: equal to the oracle, unit by unit.  (Seeds found by
    tools/translate_fuzz.py; seed 33 holds a unit that the library used to refuse with PM_ST_OFFSET_ORDER.)"""
    from test_translate_gpu import MODES, corrupt_tables
    rng = np.random.default_rng(seed)
    kw = dict(MODES[mode])
    # the same draws as tools/translate_fuzz.py makes for this seed
    assert sorted(MODES)[int(rng.integers(0, len(MODES)))] == mode
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
    check_against_oracle(harness, t, narrow)


@pytest.mark.parametrize("narrow", WIDTHS)
def test_column_gapped_in_both_rows_prints_what_the_reference_prints(narrow, harness, oracle_build):
    """A delta entry with the same column gapped in its reference row AND its query row cannot come out of a delta file
    (m_delta.cc:50-68 hands out columns once); given one, the reference's writer takes the query gap first (ties go to the query
    list) and then prints the reference gap at distance 0 -- a stray 0 inside the entry's offsets.  Same here."""
    from paramugsy_amd.translate import Tables
    i64 = lambda *v: np.array(v, dtype=np.int64)
    left = {"start": i64(1), "end": i64(100), "length": i64(100), "gap_off": i64(0, 0), "gap_start": i64(), "gap_end": i64()}
    right = {"start": i64(1), "end": i64(100), "length": i64(100), "gap_off": i64(0, 0), "gap_start": i64(), "gap_end": i64()}
    deltas = {"ref_start": i64(1), "ref_end": i64(99), "qry_start": i64(1), "qry_end": i64(99),
              "ref_gap_off": i64(0, 1), "ref_gap_start": i64(50), "ref_gap_end": i64(50),
              "qry_gap_off": i64(0, 1), "qry_gap_start": i64(50), "qry_gap_end": i64(50)}
    z = np.zeros(1, dtype=np.int32)
    t = Tables(left, right, deltas, {"delta": z, "left": z, "right": z})
    st = check_against_oracle(harness, t, narrow)
    assert st.tolist() == [0]
    _, _, ent, off = host_run(harness, t, narrow)
    assert 0 in off[:-1].tolist()  # the stray 0 is there


def test_int_path_near_its_entry_limit(harness, oracle_build, tmp_path):
    """Sequence coordinates just below 2^25 (the largest a job may hold and still take the int path): the int
    instantiation equals the oracle on every unit and its range check stays quiet."""
    w = synth.make_workload(str(tmp_path / "job"), 99, n_left=2, n_right=2, genome_len=30000, n_blocks=12, n_deltas=2,
                            entries_per_delta=30, mean_len=900)
    t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    shift = (1 << 25) - 40000
    for side in (t.left, t.right):
        side["start"] += shift
        side["end"] += shift
    for k in ("ref_start", "ref_end", "qry_start", "qry_end"):
        t.deltas[k] += shift
    biggest = max(int(np.abs(a).max()) for a in (t.left["start"], t.left["end"], t.right["start"], t.right["end"]))
    assert (1 << 24) < biggest < (1 << 25)
    for narrow in WIDTHS:
        st = check_against_oracle(harness, t, narrow)
        assert st is not None and (st == 0).all()


@pytest.mark.parametrize("narrow", WIDTHS)
def test_dropped_segment_with_more_gaps_than_the_unit_has_offsets(narrow, harness, oracle_build):
    """The FIX pass's scratch list holds (offsets of the unit + 1) gaps: enough for every segment that commits, because a gap
    owns at least one of its segment's offsets.  A segment that is dropped (b_finish does not commit) can hold more: twenty
    adjacent one-column gaps in the entry's reference row right where the left row has its own gap, and a column gapped in
    both rows further on that sends the unit to the FIX pass.  The recording stops at the cap (under ASAN, PM_HARNESS_SANITIZE=1,
    an unbounded list wrote past the scratch here) and the unit's output equals the oracle's."""
    from paramugsy_amd.translate import Tables
    i64 = lambda *v: np.array(v, dtype=np.int64)
    left = {"start": i64(1), "end": i64(199), "length": i64(200), "gap_off": i64(0, 1), "gap_start": i64(41), "gap_end": i64(41)}
    right = {"start": i64(1), "end": i64(199), "length": i64(200), "gap_off": i64(0, 1), "gap_start": i64(61), "gap_end": i64(61)}
    ref_gaps = list(range(41, 61)) + [90]
    deltas = {"ref_start": i64(1), "ref_end": i64(150), "qry_start": i64(1), "qry_end": i64(170),
              "ref_gap_off": i64(0, len(ref_gaps)), "ref_gap_start": i64(*ref_gaps), "ref_gap_end": i64(*ref_gaps),
              "qry_gap_off": i64(0, 1), "qry_gap_start": i64(90), "qry_gap_end": i64(90)}
    z = np.zeros(1, dtype=np.int32)
    t = Tables(left, right, deltas, {"delta": z, "left": z, "right": z})
    check_against_oracle(harness, t, narrow)


def test_merge_work_of_every_unit(harness, oracle_build):
    """unit_host_work (the kept gaps of a unit's four lists: what its merge walks through; -1 = dropped by the filter pass or ended by its
    set-up) on a golden job: a unit with entries in the oracle's answer has a merge in front of it, and the numbers are what
    profiles/r05_translate_merge.txt was sized with (a wavefront's longest lane against its mean lane)."""
    import pyoracle
    case = os.path.join(GOLDEN, "translate_typical")
    with open(os.path.join(case, "nucmer.list")) as f:
        deltas = [os.path.join(case, ln.strip()) for ln in f if ln.strip()]
    t = Workload.load(os.path.join(case, "profiles-l"), os.path.join(case, "profiles-r"), deltas).tables()
    ls, k1 = capi.rows_struct(t.left)
    rs, k2 = capi.rows_struct(t.right)
    ds, k3 = capi.deltas_struct(t.deltas)
    us, k4 = capi.units_struct(t.units)
    work = np.full(t.n_units, -7, np.int32)
    harness.unit_host_work.argtypes = [C.c_void_p] * 5
    assert harness.unit_host_work(C.byref(ls), C.byref(rs), C.byref(ds), C.byref(us), work.ctypes.data) == 0
    ora = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
    has_entries = np.diff(ora["unit_entry_off"]) > 0
    assert (work >= -1).all() and (work[has_entries] >= 1).all() and (work > 0).sum() >= has_entries.sum() > 0
