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


def test_dp_options_layout_and_the_environment_spelling(monkeypatch):
    """pm_dp_options_t: 18 int32 + one int64 (include/paramugsy_amd.h); the PM_DP_* switches are read by the Python binding only and
    become explicit fields (the library itself reads no environment variable: no getenv of a PM_DP_ name is left in its sources)."""
    from paramugsy_amd import dp
    assert C.sizeof(dp.PmDpOptions) == 18 * 4 + 8
    header = open(os.path.join(ROOT, "include", "paramugsy_amd.h")).read()
    body = header[header.index("typedef struct pm_dp_options {"):header.index("} pm_dp_options_t;")]
    declared = re.findall(r"int(?:32|64)_t\s+([a-z0-9_]+);", body)
    assert declared == [f for f, _ in dp.PmDpOptions._fields_]
    for name in list(os.environ):
        if name.startswith("PM_DP_"):
            monkeypatch.delenv(name)
    o = dp.options_from_env()
    assert all(getattr(o, f) == 0 for f, _ in dp.PmDpOptions._fields_)
    for k, v in {"PM_DP_MODE": "bits", "PM_DP_COLS": "8", "PM_DP_WAVES": "4", "PM_DP_BAND": "0", "PM_DP_UNI": "0", "PM_DP_DOT4": "0",
                 "PM_DP_TAIL": "0", "PM_DP_SEGMENT_CELLS": "2e6", "PM_DP_TIER_MIN_PAIRS": "8"}.items():
        monkeypatch.setenv(k, v)
    o = dp.options_from_env()
    assert (o.path_mode, o.cols_per_lane, o.waves_per_pair, o.band, o.no_uniform_depth, o.int16_weights, o.full_stripes, o.segment_cells,
            o.tier_min_pairs) == (1, 8, 4, 1, 1, 1, 1, 2000000, 8)
    monkeypatch.setenv("PM_DP_MODE", "ckpt")
    monkeypatch.setenv("PM_DP_BAND", "1")
    o = dp.options_from_env()
    assert (o.path_mode, o.band) == (2, 2)
    with pytest.raises(KeyError):
        dp.options(no_such_field=1)
    csrc = os.path.join(ROOT, "paramugsy_amd", "csrc")
    for fn in os.listdir(csrc):
        text = open(os.path.join(csrc, fn)).read()
        assert not re.search(r'getenv\("PM_DP_', text), fn


def test_translate_options_layout_and_no_environment_reads_in_the_library(monkeypatch, hip_lib):
    """pm_translate_options_t: 4 int32 + 4 reserved (include/paramugsy_amd.h).  The PM_TRANSLATE_WIDE / PM_TRANSLATE_LIBRARY_SCANS /
    PM_NO_SOA / PM_TIMING names of rounds 1-4 are a spelling of the binding and of the executables' main(); the library's own sources
    read no PM_* variable at all (they were read inside entries that run on one thread per device)."""
    assert C.sizeof(capi.PmTranslateOptions) == 8 * 4
    header = open(os.path.join(ROOT, "include", "paramugsy_amd.h")).read()
    body = header[header.index("typedef struct pm_translate_options {"):header.index("} pm_translate_options_t;")]
    declared = re.findall(r"int32_t\s+([a-z0-9_]+)(?:\[\d+\])?;", body)
    assert declared == [f for f, _ in capi.PmTranslateOptions._fields_]
    for name in ("PM_TRANSLATE_WIDE", "PM_TRANSLATE_LIBRARY_SCANS", "PM_NO_SOA", "PM_TIMING"):
        monkeypatch.delenv(name, raising=False)
    o = capi.translate_options_from_env()
    assert (o.coordinate_bits, o.library_scans, o.no_side_file, o.timing) == (0, 0, 0, 0)
    monkeypatch.setenv("PM_TRANSLATE_WIDE", "1")
    monkeypatch.setenv("PM_TRANSLATE_LIBRARY_SCANS", "1")
    monkeypatch.setenv("PM_NO_SOA", "1")
    monkeypatch.setenv("PM_TIMING", "1")
    o = capi.translate_options_from_env()
    assert (o.coordinate_bits, o.library_scans, o.no_side_file, o.timing) == (64, 1, 1, 1)
    monkeypatch.setenv("PM_TRANSLATE_WIDE", "0")
    assert capi.translate_options_from_env().coordinate_bits == 0
    # the defaults: set, refused when malformed, cleared
    bad = capi.PmTranslateOptions()
    bad.coordinate_bits = 48
    assert hip_lib.pm_translate_set_default_options(C.byref(bad)) == capi.PM_E_INVALID
    assert hip_lib.pm_translate_set_default_options(C.byref(o)) == capi.PM_OK
    assert hip_lib.pm_translate_set_default_options(None) == capi.PM_OK
    # what belongs to the library (the executables' mains and the client/worker socket header are not it)
    csrc = os.path.join(ROOT, "paramugsy_amd", "csrc")
    mains = {"m_translate_main.cc", "mugsy_profiles_main.cc", "side_tools_main.cc", "serve_common.hpp"}
    for fn in sorted(os.listdir(csrc)):
        if fn in mains:
            continue
        text = open(os.path.join(csrc, fn)).read()
        assert "getenv" not in text, fn


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
