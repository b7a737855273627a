"""ctypes loader of the CPU oracle (oracle/_build/libpm_oracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by paramugsy_amd/."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from paramugsy_amd import capi  # struct layouts only  # noqa: E402

LIB_PATH = os.path.join(_HERE, "_build", "libpm_oracle.so")
REF_DIR = os.path.join(_HERE, "_ref")
BUILD_DIR = os.path.join(_HERE, "_build")
_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s missing: run `make -C oracle oracle`" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        l.pmo_translate_units.restype = C.c_void_p
        l.pmo_translate_units.argtypes = [C.POINTER(capi.PmRows), C.POINTER(capi.PmRows), C.POINTER(capi.PmDeltas), C.POINTER(capi.PmUnits)]
        l.pmo_result_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        l.pmo_result_sizes.restype = None
        l.pmo_result_fetch.argtypes = [C.c_void_p] * 5
        l.pmo_result_fetch.restype = None
        l.pmo_result_free.argtypes = [C.c_void_p]
        l.pmo_result_free.restype = None
        for n in ("pmo_profile_idx_of_seq_idx_batch", "pmo_seq_idx_of_profile_idx_batch"):
            getattr(l, n).argtypes = [C.POINTER(capi.PmRows), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            getattr(l, n).restype = None
        l.pmo_enumerate_units.restype = C.c_int64
        l.pmo_enumerate_units.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        l.pmo_translate_files.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_char_p]
        _lib = l
    return _lib


def translate_units(left, right, deltas, units):
    """Oracle run over the same flat tables the product takes -> dict(status, unit_entry_off, entries, offsets)."""
    l = lib()
    ls, k1 = capi.rows_struct(left)
    rs, k2 = capi.rows_struct(right)
    ds, k3 = capi.deltas_struct(deltas)
    us, k4 = capi.units_struct(units)
    h = l.pmo_translate_units(C.byref(ls), C.byref(rs), C.byref(ds), C.byref(us))
    ne, no = C.c_int64(), C.c_int64()
    l.pmo_result_sizes(h, C.byref(ne), C.byref(no))
    n = int(us.n)
    status = np.zeros(n, dtype=np.int32)
    ent_off = np.zeros(n + 1, dtype=np.int64)
    entries = np.zeros(ne.value, dtype=capi.ENTRY_DTYPE)
    offsets = np.zeros(no.value, dtype=np.int64)
    l.pmo_result_fetch(h, status.ctypes.data, ent_off.ctypes.data, entries.ctypes.data, offsets.ctypes.data)
    l.pmo_result_free(h)
    return {"status": status, "unit_entry_off": ent_off, "entries": entries, "offsets": offsets}


def profile_idx_of_seq_idx(rows, row, seq_idx):
    rs, keep = capi.rows_struct(rows)
    row = np.ascontiguousarray(row, dtype=np.int32)
    q = np.ascontiguousarray(seq_idx, dtype=np.int64)
    out = np.zeros(len(q), dtype=np.int64)
    st = np.zeros(len(q), dtype=np.int32)
    lib().pmo_profile_idx_of_seq_idx_batch(C.byref(rs), len(q), row.ctypes.data, q.ctypes.data, out.ctypes.data, st.ctypes.data)
    return out, st


def seq_idx_of_profile_idx(rows, row, profile_idx):
    rs, keep = capi.rows_struct(rows)
    row = np.ascontiguousarray(row, dtype=np.int32)
    q = np.ascontiguousarray(profile_idx, dtype=np.int64)
    out = np.zeros(len(q), dtype=np.int64)
    st = np.zeros(len(q), dtype=np.int32)
    lib().pmo_seq_idx_of_profile_idx_batch(C.byref(rs), len(q), row.ctypes.data, q.ctypes.data, out.ctypes.data, st.ctypes.data)
    return out, st


def enumerate_units(left_dir, right_dir, delta_paths):
    arr = (C.c_char_p * len(delta_paths))(*[p.encode() for p in delta_paths])
    n = lib().pmo_enumerate_units(left_dir.encode(), right_dir.encode(), arr, len(delta_paths), 0, None, None, None)
    d = np.zeros(n, dtype=np.int32)
    a = np.zeros(n, dtype=np.int32)
    b = np.zeros(n, dtype=np.int32)
    lib().pmo_enumerate_units(left_dir.encode(), right_dir.encode(), arr, len(delta_paths), n, d.ctypes.data, a.ctypes.data, b.ctypes.data)
    return {"delta": d, "left": a, "right": b}


def translate_files(left_dir, right_dir, delta_paths, out_path) -> int:
    arr = (C.c_char_p * len(delta_paths))(*[p.encode() for p in delta_paths])
    return lib().pmo_translate_files(left_dir.encode(), right_dir.encode(), arr, len(delta_paths), out_path.encode())


# ---------------------------------------------------------------- profile DP oracle (oracle/dp_oracle.c)
# "Parity unpinned": no reference counterpart exists for the DP.

DP_LIB_PATH = os.path.join(_HERE, "_build", "libdp_oracle.so")
_dp = None


def dp_lib() -> C.CDLL:
    global _dp
    if _dp is None:
        if not os.path.exists(DP_LIB_PATH):
            raise RuntimeError("%s missing: run `make -C oracle oracle`" % DP_LIB_PATH)
        from paramugsy_amd.dp import PmDpParams
        o = C.CDLL(DP_LIB_PATH)
        o.dp_oracle_align_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(PmDpParams), C.c_void_p,
                                            C.c_void_p, C.c_void_p]
        o.dp_oracle_score_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(PmDpParams), C.c_void_p]
        o.dp_oracle_score_batch.restype = None
        o.dp_oracle_score_of_path.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(PmDpParams), C.c_void_p, C.c_int32,
                                              C.POINTER(C.c_int32)]
        o.dp_oracle_score_of_paths.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(PmDpParams),
                                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        o.dp_oracle_score_of_paths.restype = None
        _dp = o
    return _dp


def dp_align(inputs, params):
    """(scores, per-pair op arrays, first op first) from the scalar oracle; `inputs` is a paramugsy_amd.dp.DpInputs."""
    o = dp_lib()
    n = inputs.n_pairs
    scores = np.zeros(n, dtype=np.int32)
    n_ops = np.zeros(n, dtype=np.int32)
    ops = np.zeros(int(inputs.off_a[-1] + inputs.off_b[-1]), dtype=np.uint8)
    ca = np.ascontiguousarray(inputs.cols_a)
    cb = np.ascontiguousarray(inputs.cols_b)
    oa = np.ascontiguousarray(inputs.off_a, dtype=np.int64)
    ob = np.ascontiguousarray(inputs.off_b, dtype=np.int64)
    rc = o.dp_oracle_align_batch(ca.ctypes.data, oa.ctypes.data, cb.ctypes.data, ob.ctypes.data, n, C.byref(params),
                                 scores.ctypes.data, ops.ctypes.data, n_ops.ctypes.data)
    assert rc == 0
    paths = []
    for k in range(n):
        s = int(oa[k] + ob[k])
        paths.append(ops[s:s + int(n_ops[k])].copy())
    return scores, paths


def dp_scores(inputs, params):
    o = dp_lib()
    scores = np.zeros(inputs.n_pairs, dtype=np.int32)
    ca = np.ascontiguousarray(inputs.cols_a)
    cb = np.ascontiguousarray(inputs.cols_b)
    oa = np.ascontiguousarray(inputs.off_a, dtype=np.int64)
    ob = np.ascontiguousarray(inputs.off_b, dtype=np.int64)
    o.dp_oracle_score_batch(ca.ctypes.data, oa.ctypes.data, cb.ctypes.data, ob.ctypes.data, inputs.n_pairs, C.byref(params), scores.ctypes.data)
    return scores


def dp_score_of_path(inputs, params, k, path):
    """Re-scores a path under the specification -> (rc, score); rc != 0 when the path does not span the pair."""
    o = dp_lib()
    a = np.ascontiguousarray(inputs.cols_a[inputs.off_a[k]:inputs.off_a[k + 1]])
    b = np.ascontiguousarray(inputs.cols_b[inputs.off_b[k]:inputs.off_b[k + 1]])
    p = np.ascontiguousarray(path, dtype=np.uint8)
    s = C.c_int32()
    rc = o.dp_oracle_score_of_path(a.ctypes.data, len(a), b.ctypes.data, len(b), C.byref(params), p.ctypes.data, len(p), C.byref(s))
    return rc, s.value


def host_threads() -> int:
    try:
        return max(1, min(len(os.sched_getaffinity(0)), 16))  # a one-GPU box's CPU share is 16
    except Exception:
        return max(1, min(os.cpu_count() or 1, 16))


def _slices_by_cells(inputs, n_slices, multiple=16):
    """Contiguous slices of about equal cells, cut at multiples of `multiple` pairs (dp_tuned.c runs 16 pairs of one shape per vector)."""
    n = inputs.n_pairs
    cells = np.cumsum(np.diff(inputs.off_a).astype(np.float64) * np.diff(inputs.off_b) + 1.0)
    cuts = [0]
    for k in range(1, n_slices):
        at = int(np.searchsorted(cells, cells[-1] * k / n_slices)) if n else 0
        at = min(n, (at + multiple - 1) // multiple * multiple)
        if at > cuts[-1]:
            cuts.append(at)
    if n > cuts[-1]:
        cuts.append(n)
    return list(zip(cuts[:-1], cuts[1:]))


def dp_check_batch_exhaustively(inputs, params, scores, ops, n_ops):
    """EVERY score of a batch against the tuned CPU scorer (dp_tuned.c, itself held to dp_oracle.c by tests/test_dp_oracle.py) and
    EVERY path re-scored under the specification and checked to span its pair -- on all host cores (threads: the C calls release
    the GIL and share the batch's arrays).  Returns (pairs whose score differs, pairs whose path fails) as index arrays."""
    from concurrent.futures import ThreadPoolExecutor
    from paramugsy_amd.shard import slice_pairs
    o = dp_lib()
    dp_tuned_lib()
    n = inputs.n_pairs
    ca, cb = np.ascontiguousarray(inputs.cols_a), np.ascontiguousarray(inputs.cols_b)
    oa, ob = np.ascontiguousarray(inputs.off_a, dtype=np.int64), np.ascontiguousarray(inputs.off_b, dtype=np.int64)
    scores = np.ascontiguousarray(scores, dtype=np.int32)
    ops = np.ascontiguousarray(ops, dtype=np.uint8)
    n_ops = np.ascontiguousarray(n_ops, dtype=np.int32)
    t_scores = np.zeros(n, dtype=np.int32)
    p_rc = np.zeros(n, dtype=np.int32)
    p_scores = np.zeros(n, dtype=np.int32)
    threads = host_threads()

    def one(sl):
        lo, hi = sl
        t_scores[lo:hi] = dp_scores_tuned(slice_pairs(inputs, lo, hi), params)
        o.dp_oracle_score_of_paths(ca.ctypes.data, oa.ctypes.data, cb.ctypes.data, ob.ctypes.data, lo, hi, C.byref(params), ops.ctypes.data,
                                   n_ops.ctypes.data, p_rc.ctypes.data, p_scores.ctypes.data)
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(one, _slices_by_cells(inputs, threads * 8)))
    return np.nonzero(t_scores != scores)[0], np.nonzero((p_rc != 0) | (p_scores != scores))[0]


def dp_align_pairs(inputs, params, pairs):
    """The scalar oracle's (score, path) of the given pairs, one pair per task on all host cores."""
    from concurrent.futures import ThreadPoolExecutor
    from paramugsy_amd.shard import slice_pairs
    dp_lib()

    def one(k):
        s, p = dp_align(slice_pairs(inputs, int(k), int(k) + 1), params)
        return int(s[0]), p[0]
    with ThreadPoolExecutor(host_threads()) as pool:
        return list(pool.map(one, pairs))


# ---------------------------------------------------------------- tuned CPU scorer (oracle/dp_tuned.c): bench.py's cpu_baseline.tuned
# Compiled on the machine it runs on (gcc -O3 -march=native), into a temporary directory: a -march=native object built in one
# container must not travel to another CPU.

_dp_tuned = None


def dp_tuned_lib() -> C.CDLL:
    global _dp_tuned
    if _dp_tuned is None:
        import subprocess
        import tempfile
        from paramugsy_amd.dp import PmDpParams
        out = os.environ.get("PM_DP_TUNED_SO", "")  # built by a parent process on this machine (bench.py's worker pool)
        if not (out and os.path.exists(out)):
            out = os.path.join(tempfile.mkdtemp(prefix="pm_dp_tuned_"), "libdp_tuned.so")
            subprocess.run(["gcc", "-O3", "-march=native", "-std=c11", "-fPIC", "-shared", "-o", out, os.path.join(_HERE, "dp_tuned.c")],
                           check=True, capture_output=True)
            os.environ["PM_DP_TUNED_SO"] = out
        o = C.CDLL(out)
        o.dp_tuned_score_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(PmDpParams), C.c_void_p]
        _dp_tuned = o
    return _dp_tuned


def dp_scores_tuned(inputs, params):
    o = dp_tuned_lib()
    scores = np.zeros(inputs.n_pairs, dtype=np.int32)
    ca = np.ascontiguousarray(inputs.cols_a)
    cb = np.ascontiguousarray(inputs.cols_b)
    oa = np.ascontiguousarray(inputs.off_a, dtype=np.int64)
    ob = np.ascontiguousarray(inputs.off_b, dtype=np.int64)
    rc = o.dp_tuned_score_batch(ca.ctypes.data, oa.ctypes.data, cb.ctypes.data, ob.ctypes.data, inputs.n_pairs, C.byref(params),
                                scores.ctypes.data)
    assert rc == 0
    return scores
