"""The C-ABI library loads, exports every symbol include/paramugsy_amd.h declares, and refuses to compute
without a GPU (no CPU fallback).  CPU only: no compute call is made when a device is absent."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from paramugsy_amd import capi


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "paramugsy_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip_lib):
    names = declared_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(hip_lib, n), "libparamugsy_amd.so does not export %s" % n
    assert sorted(capi.EXPORTS) == [n for n in names if n in capi.EXPORTS]
    missing_in_binding = [n for n in names if n not in capi.EXPORTS]
    assert not missing_in_binding, missing_in_binding


def test_struct_layouts_match_header():
    assert C.sizeof(capi.PmEntry) == 48 == capi.ENTRY_DTYPE.itemsize
    assert C.sizeof(capi.PmRows) == 7 * 8
    assert C.sizeof(capi.PmDeltas) == 11 * 8
    assert C.sizeof(capi.PmUnits) == 4 * 8


def test_no_cpu_fallback_without_device(hip_lib):
    if hip_lib.pm_device_count() > 0:
        pytest.skip("a HIP device is present")
    z64 = np.zeros(1, dtype=np.int64)
    rows = {"start": z64[:0], "end": z64[:0], "length": z64[:0], "gap_off": z64, "gap_start": z64[:0], "gap_end": z64[:0]}
    deltas = {k: z64[:0] for k in ("ref_start", "ref_end", "qry_start", "qry_end", "ref_gap_start", "ref_gap_end", "qry_gap_start", "qry_gap_end")}
    deltas["ref_gap_off"] = z64
    deltas["qry_gap_off"] = z64
    z32 = np.zeros(0, dtype=np.int32)
    ls, k1 = capi.rows_struct(rows)
    ds, k2 = capi.deltas_struct(deltas)
    us, k3 = capi.units_struct({"delta": z32, "left": z32, "right": z32})
    h = C.c_void_p()
    rc = hip_lib.pm_job_create(C.byref(ls), C.byref(ls), C.byref(ds), C.byref(us), 0, C.byref(h))
    assert rc == capi.PM_E_NO_DEVICE
    assert b"no CPU path" in hip_lib.pm_last_error()
    rc = hip_lib.pm_translate_files(b"/nonexistent", b"/nonexistent", None, 0, b"/tmp/x", 0)
    assert rc == capi.PM_E_NO_DEVICE
