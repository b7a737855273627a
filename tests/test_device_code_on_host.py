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
    for fn in (h.unit_host_run, h.unit_host_run_narrow):
        fn.argtypes = [C.c_void_p] * 4 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.c_int64,
                       C.c_void_p, C.c_int64]
    return h


PM_ST_NARROW = 100  # internal status of the int instantiation: "redo this job in int64"


def host_run(h, t, narrow=False):
    run = h.unit_host_run_narrow if narrow else h.unit_host_run
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
    st, eo, ent, off = host_run(h, t, narrow)
    if narrow and (st == PM_ST_NARROW).any():
        return None
    ora = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
    assert np.array_equal(st, ora["status"])
    assert np.array_equal(eo, ora["unit_entry_off"])
    assert np.array_equal(off, ora["offsets"])
    for k in ("ref_start", "ref_end", "qry_start", "qry_end", "offset_begin", "n_offsets"):
        assert np.array_equal(ent[k], ora["entries"][k]), k
    return st


@pytest.mark.parametrize("narrow", [False, True])
@pytest.mark.parametrize("name", ["typical", "gappy", "reverse", "tiny_blocks", "empty"])
def test_device_code_equals_oracle_on_golden_inputs(name, narrow, harness, oracle_build):
    case = os.path.join(GOLDEN, "translate_" + name)
    with open(os.path.join(case, "nucmer.list")) as f:
        deltas = [os.path.join(case, ln.strip()) for ln in f if ln.strip()]
    t = Workload.load(os.path.join(case, "profiles-l"), os.path.join(case, "profiles-r"), deltas).tables()
    st = check_against_oracle(harness, t, narrow)
    assert st is not None and (st == 0).all()  # sane tables never trip the int path's range check


@pytest.mark.parametrize("narrow", [False, True])
@pytest.mark.parametrize("seed", range(4242, 4250))
def test_device_code_equals_oracle_on_inconsistent_tables(seed, narrow, harness, oracle_build, tmp_path):
    from test_translate_gpu import MODES, corrupt_tables
    w = synth.make_workload(str(tmp_path / "job"), seed, **MODES["reverse" if seed % 2 == 0 else "tiny_blocks"])
    t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    corrupt_tables(t, np.random.default_rng(seed))
    st = check_against_oracle(harness, t, narrow)
    assert st is None or (st != 0).any()


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
    for narrow in (False, True):
        st = check_against_oracle(harness, t, narrow)
        assert st is not None and (st == 0).all()
