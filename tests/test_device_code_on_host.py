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
    h.unit_host_run.argtypes = [C.c_void_p] * 4 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p,
                                                   C.c_int64, C.c_void_p, C.c_int64]
    return h


def host_run(h, t):
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
    h.unit_host_run(*args, ent.ctypes.data, 0, off.ctypes.data, 0)
    ent = np.zeros(max(1, ne.value), capi.ENTRY_DTYPE)
    off = np.zeros(max(1, no.value), np.int64)
    rc = h.unit_host_run(*args, ent.ctypes.data, ne.value, off.ctypes.data, no.value)
    assert rc == 0, "count and emit passes disagree" if rc == 2 else rc
    return st, eo, ent[:ne.value], off[:no.value]


def check_against_oracle(h, t):
    import pyoracle
    st, eo, ent, off = host_run(h, t)
    ora = pyoracle.translate_units(t.left, t.right, t.deltas, t.units)
    assert np.array_equal(st, ora["status"])
    assert np.array_equal(eo, ora["unit_entry_off"])
    assert np.array_equal(off, ora["offsets"])
    for k in ("ref_start", "ref_end", "qry_start", "qry_end", "offset_begin", "n_offsets"):
        assert np.array_equal(ent[k], ora["entries"][k]), k
    return st


@pytest.mark.parametrize("name", ["typical", "gappy", "reverse", "tiny_blocks", "empty"])
def test_device_code_equals_oracle_on_golden_inputs(name, harness, oracle_build):
    case = os.path.join(GOLDEN, "translate_" + name)
    with open(os.path.join(case, "nucmer.list")) as f:
        deltas = [os.path.join(case, ln.strip()) for ln in f if ln.strip()]
    t = Workload.load(os.path.join(case, "profiles-l"), os.path.join(case, "profiles-r"), deltas).tables()
    st = check_against_oracle(harness, t)
    assert (st == 0).all()


@pytest.mark.parametrize("seed", range(4242, 4250))
def test_device_code_equals_oracle_on_inconsistent_tables(seed, harness, oracle_build, tmp_path):
    from test_translate_gpu import MODES, corrupt_tables
    w = synth.make_workload(str(tmp_path / "job"), seed, **MODES["reverse" if seed % 2 == 0 else "tiny_blocks"])
    t = Workload.load(w.left_dir, w.right_dir, w.delta_paths).tables()
    corrupt_tables(t, np.random.default_rng(seed))
    st = check_against_oracle(harness, t)
    assert (st != 0).any()
