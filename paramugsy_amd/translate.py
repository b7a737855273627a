"""Host-side mirror of the reference's interface to the translate path.

Reference entry points (lib/m_translate):
    void Para_mugsy::translate(left_dir, right_dir, nucmer_list, out_stream)   m_translate.hh:9-14
    int  main(argc, argv)   m_translate <left_dir> <right_dir> <list> <out>    m_translate_main.cc:19-46
Here:
    translate(left_dir, right_dir, nucmer_list, out_path)   same arguments, output goes to a path
    m_translate_main(argv)                                  same argv and exit behaviour
plus the batch level the GPU path adds (a job of work units resident in HBM):
    Workload.load(...)  -> tables (rows of both sides, delta entries, units)
    TranslateJob(tables).run() / .fetch()
All arithmetic runs in libparamugsy_amd.so on the GPU; nothing here computes a translation.
"""
from __future__ import annotations

import ctypes as C
import sys
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import capi


@dataclass
class Tables:
    left: Dict[str, np.ndarray]
    right: Dict[str, np.ndarray]
    deltas: Dict[str, np.ndarray]
    units: Dict[str, np.ndarray]

    @property
    def n_units(self) -> int:
        return len(self.units["delta"])


class Workload:
    """Files of one job parsed by the library's own host code (no device needed)."""

    def __init__(self, handle: C.c_void_p, parse_error: Optional[str]):
        self._h = handle
        self.parse_error = parse_error

    @classmethod
    def load(cls, left_dir: str, right_dir: str, delta_paths: Sequence[str], allow_parse_error: bool = False) -> "Workload":
        l = capi.lib()
        arr = (C.c_char_p * len(delta_paths))(*[p.encode() for p in delta_paths])
        h = C.c_void_p()
        rc = l.pm_workload_load(left_dir.encode(), right_dir.encode(), arr, len(delta_paths), C.byref(h))
        err = None
        if rc != capi.PM_OK:
            err = l.pm_last_error().decode(errors="replace")
            if not (allow_parse_error and rc == capi.PM_E_PARSE and h):
                if h:
                    l.pm_workload_destroy(h)
                raise capi.PmError(rc, err)
        return cls(h, err)

    def tables(self) -> Tables:
        l = capi.lib()
        a, b, d, u = capi.PmRows(), capi.PmRows(), capi.PmDeltas(), capi.PmUnits()
        capi.check(l.pm_workload_tables(self._h, C.byref(a), C.byref(b), C.byref(d), C.byref(u)))
        return Tables(capi.rows_to_dict(a), capi.rows_to_dict(b), capi.deltas_to_dict(d), capi.units_to_dict(u))

    def row_name(self, side: int, row: int):
        l = capi.lib()
        major, seq = C.c_char_p(), C.c_char_p()
        capi.check(l.pm_workload_row_name(self._h, side, row, C.byref(major), C.byref(seq)))
        return major.value.decode(), seq.value.decode()

    def close(self) -> None:
        if self._h:
            capi.lib().pm_workload_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class JobResult:
    status: np.ndarray          # int32 [U]   PM_ST_* per unit
    unit_entry_off: np.ndarray  # int64 [U+1]
    entries: np.ndarray         # ENTRY_DTYPE [E]
    offsets: np.ndarray         # int64 [O]
    all_ok: bool


class TranslateJob:
    """A batch of work units resident in HBM (pm_job_*)."""

    def __init__(self, tables: Tables, device: int = 0, options: Optional[capi.PmTranslateOptions] = None):
        """options: pm_translate_options_t for this job; None = what the PM_* names spell at this moment (capi.translate_options_from_env)."""
        l = capi.lib()
        opt = options if options is not None else capi.translate_options_from_env()
        ls, k1 = capi.rows_struct(tables.left)
        rs, k2 = capi.rows_struct(tables.right)
        ds, k3 = capi.deltas_struct(tables.deltas)
        us, k4 = capi.units_struct(tables.units)
        h = C.c_void_p()
        capi.check(l.pm_job_create_opt(C.byref(ls), C.byref(rs), C.byref(ds), C.byref(us), C.byref(opt), device, C.byref(h)))
        del k1, k2, k3, k4  # the job has copied everything to the device
        self._h = h
        self.n_units = tables.n_units

    @classmethod
    def from_workload(cls, workload: "Workload", device: int = 0, options: Optional[capi.PmTranslateOptions] = None) -> "TranslateJob":
        """The job of a loaded workload with its unit list made on the device (pm_job_create_from_workload)."""
        h = C.c_void_p()
        opt = options if options is not None else capi.translate_options_from_env()
        capi.check(capi.lib().pm_job_create_from_workload_opt(workload._h, C.byref(opt), device, C.byref(h)))
        job = cls.__new__(cls)
        job._h = h
        n = C.c_int64()
        capi.check(capi.lib().pm_job_units(h, C.byref(n), None, None, None))
        job.n_units = n.value
        return job

    def units(self) -> Dict[str, np.ndarray]:
        """The job's unit list as it stands on the device."""
        out = {k: np.empty(self.n_units, np.int32) for k in ("delta", "left", "right")}
        n = C.c_int64()
        capi.check(capi.lib().pm_job_units(self._h, C.byref(n), out["delta"].ctypes.data, out["left"].ctypes.data, out["right"].ctypes.data))
        return out

    def run(self, stream: int = 0) -> None:
        """One pass of the hot path (count, scan, emit), asynchronous on `stream` (a hipStream_t value)."""
        capi.check(capi.lib().pm_job_run(self._h, C.c_void_p(stream)))

    def run_profiled(self, stream: int = 0):
        """One pass with HIP events between the phases -> (ms_filter, ms_count, ms_scan, ms_emit); waits for completion."""
        f, a, b, c = C.c_float(), C.c_float(), C.c_float(), C.c_float()
        capi.check(capi.lib().pm_job_run_profiled(self._h, C.c_void_p(stream), C.byref(f), C.byref(a), C.byref(b), C.byref(c)))
        return f.value, a.value, b.value, c.value

    def sizes(self):
        ne, no = C.c_int64(), C.c_int64()
        capi.check(capi.lib().pm_job_sizes(self._h, C.byref(ne), C.byref(no)))
        return ne.value, no.value

    def algorithmic_bytes(self) -> int:
        b = C.c_int64()
        capi.check(capi.lib().pm_job_algorithmic_bytes(self._h, C.byref(b)))
        return b.value

    def kernel_bytes(self):
        """(count pass bytes, emit pass bytes, live units) -- algorithmic, for per-kernel roofline accounting."""
        a, b, n = C.c_int64(), C.c_int64(), C.c_int64()
        capi.check(capi.lib().pm_job_kernel_bytes(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def coordinate_bits(self) -> int:
        """32 when the job's kernels run on the int tables (every number in the tables below 2^25), else 64."""
        b = C.c_int()
        capi.check(capi.lib().pm_job_coordinate_bits(self._h, C.byref(b)))
        return b.value

    def position_bits(self) -> int:
        """Width the job holds sequence positions in: 64 with coordinate_bits() == 32 is the job whose positions pass 2^25 while its rows
        are short (pm_job_position_bits)."""
        b = C.c_int()
        capi.check(capi.lib().pm_job_position_bits(self._h, C.byref(b)))
        return b.value

    def fetch(self) -> JobResult:
        ne, no = self.sizes()
        status = np.zeros(self.n_units, dtype=np.int32)
        ent_off = np.zeros(self.n_units + 1, dtype=np.int64)
        entries = np.zeros(ne, dtype=capi.ENTRY_DTYPE)
        offsets = np.zeros(no, dtype=np.int64)
        rc = capi.check(capi.lib().pm_job_fetch(self._h, status.ctypes.data, ent_off.ctypes.data, entries.ctypes.data,
                                                 offsets.ctypes.data), allow=(capi.PM_E_UNIT,))
        return JobResult(status, ent_off, entries, offsets, rc == capi.PM_OK)

    def close(self) -> None:
        if self._h:
            capi.lib().pm_job_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def profile_idx_of_seq_idx(rows: Dict[str, np.ndarray], row: np.ndarray, seq_idx: np.ndarray, device: int = 0):
    """Batched M_profile conversion a3 (lib/profiles_lib/m_profile.cc:91-112) -> (profile_idx, status)."""
    rs, keep = capi.rows_struct(rows)
    row = np.ascontiguousarray(row, dtype=np.int32)
    q = np.ascontiguousarray(seq_idx, dtype=np.int64)
    out = np.zeros(len(q), dtype=np.int64)
    st = np.zeros(len(q), dtype=np.int32)
    capi.check(capi.lib().pm_rows_profile_idx_of_seq_idx_batch(C.byref(rs), len(q), row.ctypes.data, q.ctypes.data,
                                                                out.ctypes.data, st.ctypes.data, device))
    return out, st


def seq_idx_of_profile_idx(rows: Dict[str, np.ndarray], row: np.ndarray, profile_idx: np.ndarray, device: int = 0):
    """Batched M_profile conversion a4 (lib/profiles_lib/m_profile.cc:114-149) -> (seq_idx, status); IS_NONE = gap column."""
    rs, keep = capi.rows_struct(rows)
    row = np.ascontiguousarray(row, dtype=np.int32)
    q = np.ascontiguousarray(profile_idx, dtype=np.int64)
    out = np.zeros(len(q), dtype=np.int64)
    st = np.zeros(len(q), dtype=np.int32)
    capi.check(capi.lib().pm_rows_seq_idx_of_profile_idx_batch(C.byref(rs), len(q), row.ctypes.data, q.ctypes.data,
                                                                out.ctypes.data, st.ctypes.data, device))
    return out, st


def translate(left_dir: str, right_dir: str, nucmer_list: Sequence[str], out_path: str, device: int = 0,
              options: Optional[capi.PmTranslateOptions] = None) -> None:
    """Para_mugsy::translate (m_translate.hh:9-14) preceded by m_translate_main.cc's two header lines."""
    arr = (C.c_char_p * len(nucmer_list))(*[p.encode() for p in nucmer_list])
    opt = options if options is not None else capi.translate_options_from_env()
    dev = (C.c_int32 * 1)(device)
    capi.check(capi.lib().pm_translate_files_opt(left_dir.encode(), right_dir.encode(), arr, len(nucmer_list), out_path.encode(),
                                                 left_dir.encode(), right_dir.encode(), dev, 1, C.byref(opt)))


def translate_multi(left_dir: str, right_dir: str, nucmer_list: Sequence[str], out_path: str, devices: Sequence[int]) -> None:
    """The same over a device list (pm_translate_files_multi): one host thread and one HIP context per device, the delta-file
    list cut into contiguous slices, texts joined in list order with the writer's header rule re-applied at the seams."""
    import numpy as np
    arr = (C.c_char_p * len(nucmer_list))(*[p.encode() for p in nucmer_list])
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    capi.check(capi.lib().pm_translate_files_multi(left_dir.encode(), right_dir.encode(), arr, len(nucmer_list), out_path.encode(),
                                                   dev.ctypes.data_as(C.POINTER(C.c_int32)), len(dev)))


def m_translate_main(argv: List[str]) -> int:
    """argv[0] = program name, as in m_translate_main.cc:19-46."""
    if len(argv) < 5:
        sys.stderr.write("Usage: m_translate <left_profile_dir> <right_profile_dir> <nucmer_file_list> <output_delta_path>\n")
        return 1
    with open(argv[3]) as f:
        paths = [ln.rstrip("\n") for ln in f]
    try:
        translate(argv[1], argv[2], paths, argv[4])
    except capi.PmError as e:
        sys.stderr.write("m_translate: %s\n" % e)
        return 134
    return 0
