"""ctypes binding of libparamugsy_amd.so (include/paramugsy_amd.h).

The shared library is built in-tree by `make lib` / `__graft_entry__.build()`.  There is no Python or CPU
implementation behind these calls: if the library is missing, or no HIP device is usable, the call raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PM_LIB_PATH: a differently configured build of the same library (tools/dp_mode_timing.py compares block geometries)
LIB_PATH = os.environ.get("PM_LIB_PATH") or os.path.join(_HERE, "libparamugsy_amd.so")

PM_OK = 0
PM_E_INVALID, PM_E_NO_DEVICE, PM_E_HIP, PM_E_IO, PM_E_PARSE, PM_E_UNIT, PM_E_MALFORMED = -1, -2, -3, -4, -5, -6, -7
(PM_ST_OK, PM_ST_SEQ_IDX_OUT_OF_RANGE, PM_ST_PROFILE_IDX_OUT_OF_RANGE, PM_ST_IS_NONE, PM_ST_ASSERT_GAP_BEHIND,
 PM_ST_ASSERT_SUB_LENGTHS, PM_ST_ALREADY_UNNEXT, PM_ST_STEP_LIMIT, PM_ST_OFFSET_ORDER, PM_ST_MALFORMED_INPUT,
 PM_ST_TEXT_RANGE) = range(11)

_i64p = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)


class PmRows(C.Structure):
    _fields_ = [("n", C.c_int64), ("start", _i64p), ("end", _i64p), ("length", _i64p), ("gap_off", _i64p),
                ("gap_start", _i64p), ("gap_end", _i64p)]


class PmDeltas(C.Structure):
    _fields_ = [("n", C.c_int64), ("ref_start", _i64p), ("ref_end", _i64p), ("qry_start", _i64p), ("qry_end", _i64p),
                ("ref_gap_off", _i64p), ("ref_gap_start", _i64p), ("ref_gap_end", _i64p),
                ("qry_gap_off", _i64p), ("qry_gap_start", _i64p), ("qry_gap_end", _i64p)]


class PmUnits(C.Structure):
    _fields_ = [("n", C.c_int64), ("delta", _i32p), ("left", _i32p), ("right", _i32p)]


class PmEntry(C.Structure):
    _fields_ = [("ref_start", C.c_int64), ("ref_end", C.c_int64), ("qry_start", C.c_int64), ("qry_end", C.c_int64),
                ("offset_begin", C.c_int64), ("n_offsets", C.c_int64)]


class PmTranslateOptions(C.Structure):
    """pm_translate_options_t: all zero = chosen by the library."""
    _fields_ = [("coordinate_bits", C.c_int32), ("library_scans", C.c_int32), ("no_side_file", C.c_int32), ("timing", C.c_int32),
                ("reserved", C.c_int32 * 4)]


def translate_options_from_env() -> PmTranslateOptions:
    """The PM_* names of rounds 1-4 as a SPELLING: the library reads no environment variable (include/paramugsy_amd.h,
    pm_translate_options_t); the binding, the tests and the tools read these and hand it a struct."""
    o = PmTranslateOptions()
    o.coordinate_bits = 64 if os.environ.get("PM_TRANSLATE_WIDE", "0")[:1] == "1" else 0
    o.library_scans = 1 if os.environ.get("PM_TRANSLATE_LIBRARY_SCANS", "0")[:1] == "1" else 0
    o.no_side_file = 1 if os.environ.get("PM_NO_SOA") else 0
    o.timing = 1 if os.environ.get("PM_TIMING") else 0
    return o


ENTRY_DTYPE = np.dtype([("ref_start", "<i8"), ("ref_end", "<i8"), ("qry_start", "<i8"), ("qry_end", "<i8"),
                        ("offset_begin", "<i8"), ("n_offsets", "<i8")])

# every symbol include/paramugsy_amd.h declares (tests check that the library exports each of them)
EXPORTS = [
    "pm_last_error", "pm_release_caches", "pm_device_count", "pm_device_info",
    "pm_translate_set_default_options", "pm_job_create_opt", "pm_job_create_from_workload_opt", "pm_translate_files_opt",
    "pm_job_create", "pm_job_text", "pm_job_text_fetch", "pm_job_text_fetch_range", "pm_job_run", "pm_job_run_profiled", "pm_job_sizes", "pm_job_fetch", "pm_job_algorithmic_bytes", "pm_job_kernel_bytes", "pm_job_coordinate_bits", "pm_job_position_bits", "pm_job_destroy",
    "pm_rows_profile_idx_of_seq_idx_batch", "pm_rows_seq_idx_of_profile_idx_batch",
    "pm_workload_load", "pm_workload_tables", "pm_workload_row_name", "pm_workload_destroy", "pm_job_create_from_workload", "pm_job_units",
    "pm_translate_files", "pm_translate_files_as", "pm_sort_delta", "pm_maf_analyzer", "pm_profiles_make", "pm_stage_files", "pm_untranslate",
    "pm_dp_set_default_options", "pm_dp_batch_create", "pm_dp_batch_create_opt", "pm_dp_batch_run", "pm_dp_batch_run_profiled", "pm_dp_batch_fill_busy_ms", "pm_dp_batch_fetch", "pm_dp_batch_info", "pm_dp_batch_chunks", "pm_dp_batch_variant", "pm_dp_batch_geometry", "pm_dp_batch_path_mode", "pm_dp_batch_destroy",
    "pm_dp_host_alloc", "pm_dp_host_free", "pm_dp_stream_create", "pm_dp_stream_create_opt", "pm_dp_stream_align", "pm_dp_stream_align_text", "pm_dp_stream_destroy",
    "pm_dp_pack_maf", "pm_dp_emit_maf", "pm_dp_align_maf", "pm_dp_align_blocks",
    "pm_partition", "pm_partition_weighted", "pm_delta_join_files", "pm_translate_files_multi", "pm_dp_align_multi", "pm_dp_align_blocks_multi", "pm_dp_align_maf_multi",
]


class PmError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libparamugsy_amd error %d: %s" % (code, msg))
        self.code = code


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load the HIP library; fail loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: run `make lib` (or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        l.pm_last_error.restype = C.c_char_p
        l.pm_device_count.restype = C.c_int
        l.pm_device_info.argtypes = [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), _i64p]
        l.pm_job_create.argtypes = [C.POINTER(PmRows), C.POINTER(PmRows), C.POINTER(PmDeltas), C.POINTER(PmUnits), C.c_int,
                                    C.POINTER(C.c_void_p)]
        l.pm_job_create_opt.argtypes = [C.POINTER(PmRows), C.POINTER(PmRows), C.POINTER(PmDeltas), C.POINTER(PmUnits),
                                        C.POINTER(PmTranslateOptions), C.c_int, C.POINTER(C.c_void_p)]
        l.pm_translate_set_default_options.argtypes = [C.POINTER(PmTranslateOptions)]
        l.pm_job_create_from_workload_opt.argtypes = [C.c_void_p, C.POINTER(PmTranslateOptions), C.c_int, C.POINTER(C.c_void_p)]
        l.pm_translate_files_opt.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_char_p, C.c_char_p,
                                             _i32p, C.c_int, C.POINTER(PmTranslateOptions)]
        l.pm_job_run.argtypes = [C.c_void_p, C.c_void_p]
        l.pm_job_run_profiled.argtypes = [C.c_void_p, C.c_void_p] + [C.POINTER(C.c_float)] * 4
        l.pm_job_sizes.argtypes = [C.c_void_p, _i64p, _i64p]
        l.pm_job_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.pm_job_text.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), _i64p, _i64p, _i32p]
        l.pm_job_text_fetch.argtypes = [C.c_void_p, C.c_void_p]
        l.pm_job_text_fetch_range.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
        l.pm_job_algorithmic_bytes.argtypes = [C.c_void_p, _i64p]
        l.pm_job_kernel_bytes.argtypes = [C.c_void_p, _i64p, _i64p, _i64p]
        l.pm_job_coordinate_bits.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        l.pm_job_position_bits.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        l.pm_job_destroy.argtypes = [C.c_void_p]
        l.pm_job_destroy.restype = None
        for name in ("pm_rows_profile_idx_of_seq_idx_batch", "pm_rows_seq_idx_of_profile_idx_batch"):
            getattr(l, name).argtypes = [C.POINTER(PmRows), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        l.pm_workload_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.POINTER(C.c_void_p)]
        l.pm_workload_tables.argtypes = [C.c_void_p, C.POINTER(PmRows), C.POINTER(PmRows), C.POINTER(PmDeltas), C.POINTER(PmUnits)]
        l.pm_workload_row_name.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
        l.pm_workload_destroy.argtypes = [C.c_void_p]
        l.pm_job_create_from_workload.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        l.pm_job_units.argtypes = [C.c_void_p, _i64p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.pm_workload_destroy.restype = None
        l.pm_translate_files.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_int]
        l.pm_sort_delta.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        l.pm_maf_analyzer.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        l.pm_profiles_make.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int]
        l.pm_untranslate.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p, C.c_char_p, C.c_int]
        l.pm_delta_join_files.argtypes = [C.POINTER(C.c_char_p), C.c_int, C.c_char_p]
        l.pm_partition.argtypes = [C.c_int64, C.c_int, C.c_int, _i64p, _i64p]
        l.pm_translate_files_multi.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, _i32p, C.c_int]
        # the entries that make their jobs themselves (pm_stage_files, pm_dp_align_maf, ...) take the process's defaults: the PM_*
        # spelling of this process's environment, once, when the library is loaded
        env = translate_options_from_env()
        if env.coordinate_bits or env.library_scans or env.no_side_file or env.timing:
            l.pm_translate_set_default_options(C.byref(env))
        _lib = l
    return _lib


def check(rc: int, allow: Sequence[int] = ()) -> int:
    if rc != PM_OK and rc not in allow:
        raise PmError(rc, lib().pm_last_error().decode(errors="replace"))
    return rc


def _p64(a: np.ndarray):
    return a.ctypes.data_as(_i64p)


def _p32(a: np.ndarray):
    return a.ctypes.data_as(_i32p)


def _c64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64)


def _c32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def rows_struct(rows: Dict[str, np.ndarray]) -> Tuple[PmRows, list]:
    """Dict with start,end,length,gap_off,gap_start,gap_end -> (struct, arrays to keep alive)."""
    keep = [_c64(rows[k]) for k in ("start", "end", "length", "gap_off", "gap_start", "gap_end")]
    s = PmRows(len(keep[0]), *[_p64(a) for a in keep])
    return s, keep


def deltas_struct(d: Dict[str, np.ndarray]) -> Tuple[PmDeltas, list]:
    names = ("ref_start", "ref_end", "qry_start", "qry_end", "ref_gap_off", "ref_gap_start", "ref_gap_end",
             "qry_gap_off", "qry_gap_start", "qry_gap_end")
    keep = [_c64(d[k]) for k in names]
    s = PmDeltas(len(keep[0]), *[_p64(a) for a in keep])
    return s, keep


def units_struct(u: Dict[str, np.ndarray]) -> Tuple[PmUnits, list]:
    keep = [_c32(u[k]) for k in ("delta", "left", "right")]
    s = PmUnits(len(keep[0]), *[_p32(a) for a in keep])
    return s, keep


def _copy64(ptr, n: int) -> np.ndarray:
    if n <= 0:
        return np.zeros(0, dtype=np.int64)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


def _copy32(ptr, n: int) -> np.ndarray:
    if n <= 0:
        return np.zeros(0, dtype=np.int32)
    return np.ctypeslib.as_array(ptr, shape=(n,)).copy()


def rows_to_dict(s: PmRows) -> Dict[str, np.ndarray]:
    n = int(s.n)
    off = _copy64(s.gap_off, n + 1)
    g = int(off[-1]) if n >= 0 and len(off) else 0
    return {"start": _copy64(s.start, n), "end": _copy64(s.end, n), "length": _copy64(s.length, n), "gap_off": off,
            "gap_start": _copy64(s.gap_start, g), "gap_end": _copy64(s.gap_end, g)}


def deltas_to_dict(s: PmDeltas) -> Dict[str, np.ndarray]:
    n = int(s.n)
    ro = _copy64(s.ref_gap_off, n + 1)
    qo = _copy64(s.qry_gap_off, n + 1)
    gr, gq = int(ro[-1]), int(qo[-1])
    return {"ref_start": _copy64(s.ref_start, n), "ref_end": _copy64(s.ref_end, n), "qry_start": _copy64(s.qry_start, n),
            "qry_end": _copy64(s.qry_end, n), "ref_gap_off": ro, "ref_gap_start": _copy64(s.ref_gap_start, gr),
            "ref_gap_end": _copy64(s.ref_gap_end, gr), "qry_gap_off": qo, "qry_gap_start": _copy64(s.qry_gap_start, gq),
            "qry_gap_end": _copy64(s.qry_gap_end, gq)}


def units_to_dict(s: PmUnits) -> Dict[str, np.ndarray]:
    n = int(s.n)
    return {"delta": _copy32(s.delta, n), "left": _copy32(s.left, n), "right": _copy32(s.right, n)}
