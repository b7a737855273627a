"""Profile x profile affine-gap DP on the GPU (BASELINE.json's GCUPS metric): host-side wrapper.

NO REFERENCE COUNTERPART.  orbitz/paramugsy has no DP, no scores and no traceback; the
computation is specified by this repo (oracle/dp_oracle.h) and the HIP kernel is checked against this repo's
own scalar oracle.  Everything here is plumbing around pm_dp_batch_* in libparamugsy_amd.so.
"""
from __future__ import annotations

import ctypes as C
import os
import time
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from . import capi

SYMBOLS = b"ACGT-"
OP_M, OP_I, OP_D = 0, 1, 2


class PmDpParams(C.Structure):
    _fields_ = [("sub", C.c_int32 * 25), ("gap_open", C.c_int32), ("gap_extend", C.c_int32)]


class PmDpOptions(C.Structure):
    """pm_dp_options_t (include/paramugsy_amd.h): every field 0 = chosen per batch by the library."""
    _fields_ = [("path_mode", C.c_int32), ("cols_per_lane", C.c_int32), ("waves_per_pair", C.c_int32), ("groups_per_pair", C.c_int32),
                ("band", C.c_int32), ("walk_lanes", C.c_int32), ("int16_weights", C.c_int32), ("no_uniform_depth", C.c_int32),
                ("keep_order", C.c_int32), ("no_tiers", C.c_int32), ("tier_min_pairs", C.c_int32), ("full_stripes", C.c_int32),
                ("slots", C.c_int32), ("split", C.c_int32), ("no_gate", C.c_int32), ("tile_steps", C.c_int32), ("early_walk", C.c_int32), ("reserved", C.c_int32), ("segment_cells", C.c_int64)]


def options(**fields) -> PmDpOptions:
    """pm_dp_options_t with the given fields set, e.g. options(path_mode=1, waves_per_pair=4)."""
    o = PmDpOptions()
    for k, v in fields.items():
        if k not in dict(PmDpOptions._fields_):
            raise KeyError("pm_dp_options_t has no field %r" % k)
        setattr(o, k, int(v))
    return o


# The PM_DP_* switches of earlier rounds, kept for the tests and the tools as a way to SPELL options -- read here, in the Python
# binding, and handed to the library as an explicit pm_dp_options_t; the library itself reads no environment variable.
_ENV_FIELDS = {
    "PM_DP_MODE": ("path_mode", {"bits": 1, "ckpt": 2}), "PM_DP_COLS": ("cols_per_lane", None), "PM_DP_WAVES": ("waves_per_pair", None),
    "PM_DP_GROUPS": ("groups_per_pair", None), "PM_DP_BAND": ("band", {"0": 1, "1": 2}), "PM_DP_WALK_LANES": ("walk_lanes", None),
    "PM_DP_DOT4": ("int16_weights", {"0": 1, "1": 0}), "PM_DP_UNI": ("no_uniform_depth", {"0": 1, "1": 0}), "PM_DP_KEEP_ORDER": ("keep_order", None),
    "PM_DP_NO_TIERS": ("no_tiers", None), "PM_DP_TIER_MIN_PAIRS": ("tier_min_pairs", None), "PM_DP_TAIL": ("full_stripes", {"0": 1, "1": 0}),
    "PM_DP_SLOTS": ("slots", None), "PM_DP_SPLIT": ("split", None), "PM_DP_NO_GATE": ("no_gate", None),
    "PM_DP_SEGMENT_CELLS": ("segment_cells", None), "PM_DP_TILE": ("tile_steps", {"0": 1, "*": None}),
    "PM_DP_EARLY_WALK": ("early_walk", {"0": 1, "1": 2}),  # the walk beside the fill kernel of its own launch: never / wherever possible  # PM_DP_TILE=0: never; =N: tiles of N steps
}


def options_from_env(env=None) -> PmDpOptions:
    env = os.environ if env is None else env
    o = PmDpOptions()
    for name, (field, table) in _ENV_FIELDS.items():
        v = env.get(name)
        if v is None or v == "":
            continue
        if table is not None and "*" in table and v not in table:
            setattr(o, field, int(float(v)))  # a number that is not one of the table's words goes through as it is
        else:
            setattr(o, field, table.get(v, 0) if table is not None else int(float(v)))
    return o


def set_default_options(opt: PmDpOptions = None) -> None:
    """The options of batches the library makes itself (pm_dp_align_*, pm_dp_stream_create without options); None: all chosen."""
    capi.check(_lib().pm_dp_set_default_options(C.byref(opt) if opt is not None else None))


def make_params(rows_a: int, rows_b: int, match: int = 5, mismatch: int = -4, base_gap: int = -3,
                open_per_pair: int = 8, extend_per_pair: int = 2) -> PmDpParams:
    """Sum-of-pairs scoring: 5x5 matrix over A,C,G,T,gap and gap penalties scaled by the number of row pairs."""
    p = PmDpParams()
    for a in range(5):
        for b in range(5):
            if a == 4 and b == 4:
                v = 0
            elif a == 4 or b == 4:
                v = base_gap
            else:
                v = match if a == b else mismatch
            p.sub[a * 5 + b] = v
    p.gap_open = open_per_pair * rows_a * rows_b
    p.gap_extend = extend_per_pair * rows_a * rows_b
    return p


def pack_profile(rows: Sequence[bytes]) -> np.ndarray:
    """Gapped row texts of one alignment block -> uint8 [columns, 8] packed columns {nA,nC,nG,nT,nGap,nOther,0,0}.

    Case-insensitive.  A symbol that is neither ACGT nor '-' (N, IUPAC codes) is counted in byte 5, which the DP does not
    score: such a row is neutral in that column, and every column's six counts sum to the number of rows.  Host-side
    helper for tests; the product's entry point for MAF blocks is pm_dp_pack_maf."""
    if len(rows) > 255:
        raise ValueError("a packed column counts rows in a byte: at most 255 rows, got %d" % len(rows))
    n = len(rows[0])
    out = np.zeros((n, 8), dtype=np.uint8)
    for r in rows:
        a = np.frombuffer(r.upper(), dtype=np.uint8)
        if len(a) != n:
            raise ValueError("rows of one block must have the same number of columns")
        known = np.zeros(n, dtype=bool)
        for k, ch in enumerate(SYMBOLS):
            hit = a == ch
            out[:, k] += hit.astype(np.uint8)
            known |= hit
        out[:, 5] += (~known).astype(np.uint8)
    return out


def _lib():
    l = capi.lib()
    if not getattr(l, "_dp_bound", False):
        l.pm_dp_batch_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(PmDpParams), C.c_int64,
                                         C.c_int, C.POINTER(C.c_void_p)]
        l.pm_dp_batch_create_opt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(PmDpParams),
                                             C.POINTER(PmDpOptions), C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        l.pm_dp_set_default_options.argtypes = [C.POINTER(PmDpOptions)]
        l.pm_dp_stream_create_opt.argtypes = [C.POINTER(PmDpParams), C.POINTER(PmDpOptions), C.c_int32, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        l.pm_dp_batch_run.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        l.pm_dp_batch_run_profiled.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.pm_dp_batch_fill_busy_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        l.pm_dp_batch_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        l.pm_dp_batch_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
        l.pm_dp_batch_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
        l.pm_dp_batch_variant.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.pm_dp_batch_chunks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        l.pm_dp_batch_path_mode.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        l.pm_dp_pack_maf.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        l.pm_dp_emit_maf.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        l.pm_dp_align_maf.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(PmDpParams), C.c_char_p, C.c_int]
        l.pm_dp_align_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                         C.POINTER(PmDpParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int]
        l.pm_dp_align_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(PmDpParams), C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p]
        l.pm_dp_align_blocks_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                               C.POINTER(PmDpParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        l.pm_dp_align_maf_multi.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(PmDpParams), C.c_char_p, C.c_void_p, C.c_int]
        l.pm_dp_batch_destroy.argtypes = [C.c_void_p]
        l.pm_dp_batch_destroy.restype = None
        l.pm_dp_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_int64]
        l.pm_dp_host_free.argtypes = [C.c_void_p]
        l.pm_dp_host_free.restype = None
        l.pm_dp_stream_create.argtypes = [C.POINTER(PmDpParams), C.c_int32, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]
        l.pm_dp_stream_align.argtypes = [C.c_void_p] * 5 + [C.c_int64] + [C.c_void_p] * 3
        l.pm_dp_stream_align_text.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                              C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        l.pm_dp_stream_destroy.argtypes = [C.c_void_p]
        l.pm_dp_stream_destroy.restype = None
        l._dp_bound = True
    return l


@dataclass
class DpInputs:
    cols_a: np.ndarray  # uint8 [total_a, 8]
    off_a: np.ndarray   # int64 [n+1]
    cols_b: np.ndarray
    off_b: np.ndarray

    @property
    def n_pairs(self) -> int:
        return len(self.off_a) - 1

    @property
    def cells(self) -> int:
        return int((np.diff(self.off_a) * np.diff(self.off_b)).sum())


class DpBatch:
    """A batch of profile pairs resident in HBM (pm_dp_batch_*)."""

    def __init__(self, inputs: DpInputs, params: PmDpParams, device: int = 0, tb_budget_bytes: int = 0, options: PmDpOptions = None):
        """options None: what the PM_DP_* environment spells (options_from_env; nothing set = everything chosen by the library)."""
        l = _lib()
        opt = options if options is not None else options_from_env()
        self.inputs = inputs
        ca = np.ascontiguousarray(inputs.cols_a, dtype=np.uint8)
        cb = np.ascontiguousarray(inputs.cols_b, dtype=np.uint8)
        oa = np.ascontiguousarray(inputs.off_a, dtype=np.int64)
        ob = np.ascontiguousarray(inputs.off_b, dtype=np.int64)
        h = C.c_void_p()
        capi.check(l.pm_dp_batch_create_opt(ca.ctypes.data, oa.ctypes.data, cb.ctypes.data, ob.ctypes.data, len(oa) - 1, C.byref(params),
                                            C.byref(opt), tb_budget_bytes, device, C.byref(h)))
        self._h = h
        self._oa, self._ob = oa, ob

    def run(self, traceback: bool = True, stream: int = 0) -> None:
        capi.check(_lib().pm_dp_batch_run(self._h, 1 if traceback else 0, C.c_void_p(stream)))

    def run_profiled(self, traceback: bool = True, stream: int = 0) -> Tuple[float, float]:
        a, b = C.c_float(), C.c_float()
        capi.check(_lib().pm_dp_batch_run_profiled(self._h, 1 if traceback else 0, C.c_void_p(stream), C.byref(a), C.byref(b)))
        return a.value, b.value

    def fill_busy_ms(self) -> float:
        """After run_profiled: the time during which some fill kernel ran (the launches of a batch of several chunks overlap)."""
        v = C.c_float()
        capi.check(_lib().pm_dp_batch_fill_busy_ms(self._h, C.byref(v)))
        return v.value

    def info(self):
        cells, tb, inp, chunks = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32()
        capi.check(_lib().pm_dp_batch_info(self._h, C.byref(cells), C.byref(tb), C.byref(inp), C.byref(chunks)))
        return {"cells": cells.value, "traceback_bytes": tb.value, "input_bytes": inp.value, "chunks": chunks.value}

    def chunks(self):
        """(first position of every chunk plus n_pairs, processing order): chunk c handles pairs order[first[c]:first[c+1]]."""
        n = self.info()["chunks"]
        first = np.zeros(n + 1, dtype=np.int64)
        order = np.zeros(len(self._oa) - 1, dtype=np.int32)
        capi.check(_lib().pm_dp_batch_chunks(self._h, first.ctypes.data, n + 1, order.ctypes.data))
        return first, order

    def geometry(self):
        """{"padded_cells": the cells the fill kernel's stripes cover, "narrow_last_stripes": bool, "fill_launches": per pass}."""
        c, t, n = C.c_int64(), C.c_int32(), C.c_int64()
        capi.check(_lib().pm_dp_batch_geometry(self._h, C.byref(c), C.byref(t), C.byref(n)))
        return {"padded_cells": c.value, "narrow_last_stripes": bool(t.value), "fill_launches": n.value}

    def variant(self):
        a, b, c = C.c_int32(), C.c_int32(), C.c_int32()
        capi.check(_lib().pm_dp_batch_variant(self._h, C.byref(a), C.byref(b), C.byref(c)))
        ck, br, bc = C.c_int32(), C.c_int32(), C.c_int32()
        capi.check(_lib().pm_dp_batch_path_mode(self._h, C.byref(ck), C.byref(br), C.byref(bc)))
        return {"cols_per_lane": a.value, "dot4": bool(b.value & 1), "uniform_depth": bool(b.value & 2), "valu_ops_per_cell": c.value,
                "checkpoints": bool(ck.value),
                "block_rows": br.value, "block_columns": bc.value}

    def fetch(self, with_paths: bool = True):
        n = len(self._oa) - 1
        scores = np.zeros(n, dtype=np.int32)
        n_ops = np.zeros(n, dtype=np.int32)
        ops = np.zeros(int(self._oa[-1] + self._ob[-1]) if with_paths else 0, dtype=np.uint8)
        capi.check(_lib().pm_dp_batch_fetch(self._h, scores.ctypes.data, ops.ctypes.data if with_paths else None,
                                            n_ops.ctypes.data if with_paths else None))
        return scores, ops, n_ops

    def paths(self, ops: np.ndarray, n_ops: np.ndarray) -> List[np.ndarray]:
        """Per-pair op arrays (first op first) from the right-aligned slots."""
        out = []
        for k in range(len(n_ops)):
            end = int(self._oa[k + 1] + self._ob[k + 1])
            out.append(ops[end - int(n_ops[k]):end])
        return out

    def close(self) -> None:
        if self._h:
            _lib().pm_dp_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---------------------------------------------------------------- the DP fed from host memory (pm_dp_stream_*)

class _PinnedAllocation:
    """Frees a pm_dp_host_alloc allocation when the last reference to it goes."""

    def __init__(self, ptr):
        self._p = ptr

    def __del__(self):
        try:
            if self._p:
                _lib().pm_dp_host_free(self._p)
                self._p = None
        except Exception:
            pass


class PinnedArray:
    """A numpy array over pinned host memory from pm_dp_host_alloc (copies from / to it are asynchronous).

    `.a` OWNS the allocation: the memory is freed when the last array that views it is gone (the ctypes buffer at the bottom of
    every view's `.base` chain holds the allocation), not when this object is -- round 3 freed it with the PinnedArray, and a
    copy engine that was still reading `.a` of a collected PinnedArray took a GPU memory access fault.  close() only drops this
    object's own reference."""

    def __init__(self, shape, dtype):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        capi.check(_lib().pm_dp_host_alloc(C.byref(p), n))
        buf = (C.c_uint8 * max(n, 1)).from_address(p.value)
        buf._pm_allocation = _PinnedAllocation(p)  # lives as long as the buffer, i.e. as long as any view of it
        self.a = np.frombuffer(buf, dtype=np.uint8, count=n).view(dtype).reshape(shape)

    def close(self):
        self.a = None

    def __del__(self):
        self.a = None


class DpStream:
    """A batch from host memory: uploaded in segments while the fill kernel already runs on those that have arrived; results
    back on a third stream (pm_dp_stream_*)."""

    def __init__(self, params: PmDpParams, segments: int = 4, workspace_bytes: int = 0, device: int = 0, options: PmDpOptions = None):
        h = C.c_void_p()
        opt = options if options is not None else options_from_env()
        capi.check(_lib().pm_dp_stream_create_opt(C.byref(params), C.byref(opt), segments, workspace_bytes, device, C.byref(h)))
        self._h = h

    def align(self, inputs: DpInputs, scores: np.ndarray = None, ops: np.ndarray = None, n_ops: np.ndarray = None, with_paths: bool = True):
        """Returns (scores, ops, n_ops) in pm_dp_batch_fetch's layout; pass pinned output arrays to keep the downloads asynchronous."""
        n = inputs.n_pairs
        oa = np.ascontiguousarray(inputs.off_a, dtype=np.int64)
        ob = np.ascontiguousarray(inputs.off_b, dtype=np.int64)
        ca = inputs.cols_a if inputs.cols_a.flags.c_contiguous else np.ascontiguousarray(inputs.cols_a)
        cb = inputs.cols_b if inputs.cols_b.flags.c_contiguous else np.ascontiguousarray(inputs.cols_b)
        if scores is None:
            scores = np.zeros(n, dtype=np.int32)
        if with_paths and ops is None:
            ops = np.zeros(max(1, int(oa[-1] + ob[-1])), dtype=np.uint8)
            n_ops = np.zeros(n, dtype=np.int32)
        capi.check(_lib().pm_dp_stream_align(self._h, ca.ctypes.data, oa.ctypes.data, cb.ctypes.data, ob.ctypes.data, n, scores.ctypes.data,
                                             ops.ctypes.data if with_paths else None, n_ops.ctypes.data if with_paths else None))
        return scores, ops, n_ops

    def align_text(self, side_a, side_b, scores: np.ndarray = None, ops: np.ndarray = None, n_ops: np.ndarray = None, with_paths: bool = True):
        """Row texts in (pm_dp_stream_align_text): side = (text uint8, row_off int64, block_row int64), the flat description of
        flatten_blocks; packed on the device segment by segment.  Same outputs as align()."""
        ta, roa, bra = side_a
        tb, rob, brb = side_b
        n = len(bra) - 1
        cols = lambda ro, br: sum(int(ro[br[k] + 1] - ro[br[k]]) for k in range(n) if br[k + 1] > br[k])  # noqa: E731
        if scores is None:
            scores = np.zeros(max(1, n), dtype=np.int32)
        if with_paths and ops is None:
            ops = np.zeros(max(1, cols(roa, bra) + cols(rob, brb)), dtype=np.uint8)
            n_ops = np.zeros(max(1, n), dtype=np.int32)
        capi.check(_lib().pm_dp_stream_align_text(self._h, ta.ctypes.data, roa.ctypes.data, len(roa) - 1, bra.ctypes.data, tb.ctypes.data,
                                                  rob.ctypes.data, len(rob) - 1, brb.ctypes.data, n, scores.ctypes.data,
                                                  ops.ctypes.data if with_paths else None, n_ops.ctypes.data if with_paths else None))
        return scores, ops, n_ops

    def close(self) -> None:
        if self._h:
            _lib().pm_dp_stream_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def paths_of(inputs: DpInputs, ops: np.ndarray, n_ops: np.ndarray) -> List[np.ndarray]:
    """Per-pair op arrays (first op first) from the right-aligned slots of pm_dp_batch_fetch / pm_dp_stream_align."""
    out = []
    for k in range(len(n_ops)):
        end = int(inputs.off_a[k + 1] + inputs.off_b[k + 1])
        out.append(ops[end - int(n_ops[k]):end])
    return out


# ---------------------------------------------------------------- MAF blocks in and out (pm_dp_pack_maf / pm_dp_emit_maf)

def flatten_blocks(blocks: Sequence[Sequence[bytes]]):
    """A list of blocks (each a list of equally long gapped row texts) -> (text uint8, row_off int64, block_row int64), the flat
    description the C ABI takes."""
    rows = [r for b in blocks for r in b]
    text = np.frombuffer(b"".join(rows), dtype=np.uint8).copy() if rows else np.zeros(0, dtype=np.uint8)
    row_off = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    block_row = np.concatenate([[0], np.cumsum([len(b) for b in blocks])]).astype(np.int64)
    return text, row_off, block_row


def pack_maf(blocks: Sequence[Sequence[bytes]], device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Rows of every block -> (packed columns uint8 [total columns, 8], col_off int64 [n_blocks + 1]) on the GPU."""
    l = _lib()
    text, row_off, block_row = flatten_blocks(blocks)
    n = len(block_row) - 1
    col_off = np.zeros(n + 1, dtype=np.int64)
    capi.check(l.pm_dp_pack_maf(text.ctypes.data, row_off.ctypes.data, len(row_off) - 1, block_row.ctypes.data, n, None, col_off.ctypes.data, device))
    cols = np.zeros((int(col_off[-1]), 8), dtype=np.uint8)
    capi.check(l.pm_dp_pack_maf(text.ctypes.data, row_off.ctypes.data, len(row_off) - 1, block_row.ctypes.data, n, cols.ctypes.data,
                                col_off.ctypes.data, device))
    return cols, col_off


def emit_maf(blocks_a: Sequence[Sequence[bytes]], blocks_b: Sequence[Sequence[bytes]], paths: Sequence[np.ndarray],
             device: int = 0) -> List[List[bytes]]:
    """Pair k = block k of each side + its path -> the merged block's rows (A's rows first), on the GPU."""
    l = _lib()
    ta, roa, bra = flatten_blocks(blocks_a)
    tb, rob, brb = flatten_blocks(blocks_b)
    n = len(paths)
    n_ops = np.array([len(p) for p in paths], dtype=np.int32)
    ops_off = np.concatenate([[0], np.cumsum(n_ops)]).astype(np.int64)
    ops = np.concatenate([np.asarray(p, dtype=np.uint8) for p in paths]) if n and ops_off[-1] else np.zeros(1, dtype=np.uint8)
    out_off = np.zeros(n + 1, dtype=np.int64)
    args = [ta.ctypes.data, roa.ctypes.data, len(roa) - 1, bra.ctypes.data, tb.ctypes.data, rob.ctypes.data, len(rob) - 1, brb.ctypes.data, n,
            ops.ctypes.data, ops_off.ctypes.data, n_ops.ctypes.data]
    capi.check(l.pm_dp_emit_maf(*args, None, out_off.ctypes.data, device))
    out = np.zeros(max(1, int(out_off[-1])), dtype=np.uint8)
    capi.check(l.pm_dp_emit_maf(*args, out.ctypes.data, out_off.ctypes.data, device))
    merged = []
    for k in range(n):
        rows = len(blocks_a[k]) + len(blocks_b[k])
        ln = int(n_ops[k])
        base = int(out_off[k])
        merged.append([out[base + r * ln: base + (r + 1) * ln].tobytes() for r in range(rows)])
    return merged


def align_blocks(blocks_a: Sequence[Sequence[bytes]], blocks_b: Sequence[Sequence[bytes]], params: PmDpParams,
                 device: int = 0) -> Tuple[np.ndarray, List[List[bytes]]]:
    """Pair k = block k of each side -> (scores, merged blocks), pack + DP + expansion in one call (pm_dp_align_blocks): the
    texts go to the device once, the packed columns and the paths stay there."""
    ta, roa, bra = flatten_blocks(blocks_a)
    tb, rob, brb = flatten_blocks(blocks_b)
    n = len(blocks_a)
    if len(blocks_b) != n:
        raise ValueError("pair k is block k of each side: %d and %d blocks" % (n, len(blocks_b)))
    cap = sum((len(a) + len(b)) * ((len(a[0]) if a else 0) + (len(b[0]) if b else 0)) for a, b in zip(blocks_a, blocks_b))
    scores = np.zeros(max(1, n), dtype=np.int32)
    cols = np.zeros(max(1, n), dtype=np.int32)
    out = np.zeros(max(1, cap), dtype=np.uint8)
    out_off = np.zeros(n + 1, dtype=np.int64)
    capi.check(_lib().pm_dp_align_blocks(ta.ctypes.data, roa.ctypes.data, len(roa) - 1, bra.ctypes.data, tb.ctypes.data, rob.ctypes.data,
                                         len(rob) - 1, brb.ctypes.data, n, C.byref(params), scores.ctypes.data, cols.ctypes.data,
                                         out.ctypes.data, cap, out_off.ctypes.data, device))
    merged = []
    for k in range(n):
        rows, ln, base = len(blocks_a[k]) + len(blocks_b[k]), int(cols[k]), int(out_off[k])
        merged.append([out[base + r * ln: base + (r + 1) * ln].tobytes() for r in range(rows)])
    return scores[:n], merged


def align_maf_files(maf_a: str, maf_b: str, params: PmDpParams, out_maf: str, device: int = 0, devices: Sequence[int] = None) -> None:
    """devices: a device list -> pm_dp_align_maf_multi (the blocks cut into contiguous slices, one host thread per device)."""
    if devices is not None:
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        capi.check(_lib().pm_dp_align_maf_multi(maf_a.encode(), maf_b.encode(), C.byref(params), out_maf.encode(), dev.ctypes.data, len(dev)))
        return
    capi.check(_lib().pm_dp_align_maf(maf_a.encode(), maf_b.encode(), C.byref(params), out_maf.encode(), device))


def align_multi(inputs: DpInputs, params: PmDpParams, devices: Sequence[int], with_paths: bool = True):
    """pm_dp_align_multi: host columns in, (scores, ops, n_ops) out in pm_dp_batch_fetch's layout, the pairs cut into contiguous
    slices over `devices` (one host thread and one HIP context each; no torch.distributed anywhere)."""
    n = inputs.n_pairs
    oa = np.ascontiguousarray(inputs.off_a, dtype=np.int64)
    ob = np.ascontiguousarray(inputs.off_b, dtype=np.int64)
    ca = np.ascontiguousarray(inputs.cols_a, dtype=np.uint8)
    cb = np.ascontiguousarray(inputs.cols_b, dtype=np.uint8)
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    scores = np.zeros(max(1, n), dtype=np.int32)
    ops = np.zeros(max(1, int(oa[-1] + ob[-1])), dtype=np.uint8) if with_paths else None
    n_ops = np.zeros(max(1, n), dtype=np.int32) if with_paths else None
    capi.check(_lib().pm_dp_align_multi(ca.ctypes.data, oa.ctypes.data, cb.ctypes.data, ob.ctypes.data, n, C.byref(params), dev.ctypes.data, len(dev),
                                        scores.ctypes.data, ops.ctypes.data if with_paths else None, n_ops.ctypes.data if with_paths else None))
    return scores[:n], ops, (n_ops[:n] if with_paths else None)


def align_blocks_multi(blocks_a: Sequence[Sequence[bytes]], blocks_b: Sequence[Sequence[bytes]], params: PmDpParams,
                       devices: Sequence[int]) -> Tuple[np.ndarray, List[List[bytes]]]:
    """pm_dp_align_blocks_multi: align_blocks over a device list; merged blocks gathered on the host in pair order."""
    ta, roa, bra = flatten_blocks(blocks_a)
    tb, rob, brb = flatten_blocks(blocks_b)
    n = len(blocks_a)
    if len(blocks_b) != n:
        raise ValueError("pair k is block k of each side: %d and %d blocks" % (n, len(blocks_b)))
    cap = sum((len(a) + len(b)) * ((len(a[0]) if a else 0) + (len(b[0]) if b else 0)) for a, b in zip(blocks_a, blocks_b))
    scores = np.zeros(max(1, n), dtype=np.int32)
    cols = np.zeros(max(1, n), dtype=np.int32)
    out = np.zeros(max(1, cap), dtype=np.uint8)
    out_off = np.zeros(n + 1, dtype=np.int64)
    dev = np.ascontiguousarray(devices, dtype=np.int32)
    capi.check(_lib().pm_dp_align_blocks_multi(ta.ctypes.data, roa.ctypes.data, len(roa) - 1, bra.ctypes.data, tb.ctypes.data, rob.ctypes.data,
                                               len(rob) - 1, brb.ctypes.data, n, C.byref(params), dev.ctypes.data, len(dev), scores.ctypes.data,
                                               cols.ctypes.data, out.ctypes.data, cap, out_off.ctypes.data))
    merged = []
    for k in range(n):
        rows, ln, base = len(blocks_a[k]) + len(blocks_b[k]), int(cols[k]), int(out_off[k])
        merged.append([out[base + r * ln: base + (r + 1) * ln].tobytes() for r in range(rows)])
    return scores[:n], merged


# ---------------------------------------------------------------- synthetic workloads

def synth_pairs(seed: int, n_pairs: int, rows: int, length: int, sub_rate: float = 0.08, indel_rate: float = 0.01,
                row_noise: float = 0.1, gap_col_rate: float = 0.05, vary_length: bool = False) -> DpInputs:
    """`n_pairs` pairs of `rows`-row profiles, `length` columns each (PCG64, seeded).

    A's consensus is uniform ACGT; B's consensus is A's with substitutions and short indels, cut or padded to
    `length` (or, with vary_length, left at its natural length); each row copies its consensus with probability
    1-row_noise, else a random base; a column of a row is a gap with probability gap_col_rate."""
    rng = np.random.default_rng(seed)

    def rows_to_counts(cons: np.ndarray) -> np.ndarray:
        L = len(cons)
        r = np.broadcast_to(cons, (rows, L)).copy()
        noise = rng.random((rows, L)) < row_noise
        r[noise] = rng.integers(0, 4, size=int(noise.sum()))
        gaps = rng.random((rows, L)) < gap_col_rate
        r[gaps] = 4
        out = np.zeros((L, 8), dtype=np.uint8)
        for s in range(5):
            out[:, s] = (r == s).sum(axis=0)
        return out

    A, B, la, lb = [], [], [], []
    for _ in range(n_pairs):
        ca = rng.integers(0, 4, size=length)
        cb = ca.copy()
        subs = rng.random(length) < sub_rate
        cb[subs] = rng.integers(0, 4, size=int(subs.sum()))
        n_ind = rng.poisson(length * indel_rate)
        for _k in range(int(n_ind)):
            at = int(rng.integers(0, len(cb)))
            ln = int(rng.geometric(0.5))
            if rng.random() < 0.5:
                cb = np.delete(cb, slice(at, at + ln))
            else:
                cb = np.insert(cb, at, rng.integers(0, 4, size=ln))
        if not vary_length:
            if len(cb) >= length:
                cb = cb[:length]
            else:
                cb = np.concatenate([cb, rng.integers(0, 4, size=length - len(cb))])
        if len(cb) == 0:
            cb = rng.integers(0, 4, size=1)
        A.append(rows_to_counts(ca))
        B.append(rows_to_counts(cb))
        la.append(len(ca))
        lb.append(len(cb))
    off_a = np.concatenate([[0], np.cumsum(la)]).astype(np.int64)
    off_b = np.concatenate([[0], np.cumsum(lb)]).astype(np.int64)
    return DpInputs(np.concatenate(A), off_a, np.concatenate(B), off_b)


def synth_pairs_fast(seed: int, n_pairs: int, rows: int, length: int, sub_rate: float = 0.08, row_noise: float = 0.1,
                     gap_col_rate: float = 0.05, shift_rate: float = 0.3, with_rows: bool = False):
    """Vectorised generator for large benches: equal lengths; B = A's consensus with substitutions and, for a
    fraction of pairs, a cyclic shift by a few columns (so optimal paths carry gaps).
    with_rows: also return the two sides' row texts as flat block descriptions (text uint8, row_off, block_row) -- the
    rows the packed columns count, for the entries that take MAF rows (pm_dp_stream_align_text, pm_dp_align_blocks)."""
    rng = np.random.default_rng(seed)
    L = length
    ca = rng.integers(0, 4, size=(n_pairs, L), dtype=np.int8)
    cb = ca.copy()
    subs = rng.random((n_pairs, L)) < sub_rate
    cb[subs] = rng.integers(0, 4, size=int(subs.sum()), dtype=np.int8)
    shift = np.where(rng.random(n_pairs) < shift_rate, rng.integers(1, 6, size=n_pairs), 0)
    for k in np.nonzero(shift)[0]:
        cb[k] = np.roll(cb[k], int(shift[k]))
    letters = np.frombuffer(SYMBOLS, dtype=np.uint8)

    def counts(cons: np.ndarray):
        out = np.zeros((n_pairs, L, 8), dtype=np.uint8)
        text = np.zeros((n_pairs, rows, L), dtype=np.uint8) if with_rows else None
        for k in range(rows):
            r = cons.copy()
            noise = rng.random((n_pairs, L)) < row_noise
            r[noise] = rng.integers(0, 4, size=int(noise.sum()), dtype=np.int8)
            r[rng.random((n_pairs, L)) < gap_col_rate] = 4
            for s in range(5):
                out[:, :, s] += (r == s)
            if with_rows:
                text[:, k, :] = letters[r]
        return out.reshape(n_pairs * L, 8), text

    off = (np.arange(n_pairs + 1, dtype=np.int64) * L)
    cols_a, text_a = counts(ca)
    cols_b, text_b = counts(cb)
    inputs = DpInputs(cols_a, off, cols_b, off.copy())
    if not with_rows:
        return inputs
    row_off = np.arange(n_pairs * rows + 1, dtype=np.int64) * L
    block_row = np.arange(n_pairs + 1, dtype=np.int64) * rows
    return inputs, (text_a.reshape(-1), row_off, block_row), (text_b.reshape(-1), row_off.copy(), block_row.copy())


def synth_batch(seed: int, la, lb, rows_a: int, rows_b: int, sub_rate: float = 0.08, row_noise: float = 0.1,
                gap_col_rate: float = 0.05, shift_rate: float = 0.3) -> DpInputs:
    """Vectorised generator for big and for ragged batches (PCG64, seeded): pair k has la[k] x lb[k] columns.

    A's consensus is uniform ACGT.  B's consensus is A's resampled to lb[k] columns (column j of B copies column
    floor(j * la / lb) of A, so unequal lengths put evenly spread indels on the optimal path), cyclically shifted by 1-5
    columns for a fraction `shift_rate` of the pairs, with substitutions at `sub_rate`.  Every row copies its consensus with
    probability 1 - row_noise (else a random base) and is a gap with probability gap_col_rate."""
    rng = np.random.default_rng(seed)
    la = np.asarray(la, dtype=np.int64)
    lb = np.asarray(lb, dtype=np.int64)
    n = len(la)
    off_a = np.concatenate([[0], np.cumsum(la)]).astype(np.int64)
    off_b = np.concatenate([[0], np.cumsum(lb)]).astype(np.int64)
    ta, tb = int(off_a[-1]), int(off_b[-1])
    cons_a = rng.integers(0, 4, size=ta, dtype=np.uint8)
    it = np.int32 if max(ta, tb) < (1 << 30) else np.int64  # index arithmetic in 32 bits when it fits
    pid = np.repeat(np.arange(n, dtype=np.int32), lb)
    pos = np.arange(tb, dtype=it) - off_b[:-1].astype(it)[pid]
    shift = np.where(rng.random(n) < shift_rate, rng.integers(1, 6, size=n), 0).astype(it)
    lbp = np.maximum(lb, 1).astype(it)[pid]
    pos += shift[pid]
    pos -= np.where(pos >= lbp, lbp, it(0))  # cyclic shift by < 6 columns (profiles shorter than that wrap twice at most)
    pos -= np.where(pos >= lbp, lbp, it(0))
    pos = np.minimum(pos, lbp - 1)
    if np.array_equal(la, lb):
        src = off_a[:-1].astype(it)[pid] + pos
    else:
        lap = la.astype(np.int64)[pid]
        src = off_a[:-1][pid] + np.minimum((pos.astype(np.int64) * lap) // lbp, np.maximum(lap - 1, 0))
    if ta > 0:
        cons_b = cons_a[np.minimum(src, ta - 1)]
    else:
        cons_b = rng.integers(0, 4, size=tb, dtype=np.uint8)
    del pid, pos, src, lbp
    subs = rng.integers(0, 65536, size=tb, dtype=np.uint16) < int(sub_rate * 65536)
    cons_b[subs] = rng.integers(0, 4, size=int(subs.sum()), dtype=np.uint8)
    del subs

    t_noise, t_gap = np.uint32(int(row_noise * 65536)), np.uint32(int(gap_col_rate * 16384))

    def counts(cons: np.ndarray, rows: int) -> np.ndarray:
        acc = np.zeros(len(cons), dtype=np.uint64)
        for _r in range(rows):
            x = rng.integers(0, 1 << 32, size=len(cons), dtype=np.uint32)  # bits 0-15 noise, 16-17 a base, 18-31 gap
            r = np.where((x & np.uint32(0xffff)) < t_noise, ((x >> np.uint32(16)) & np.uint32(3)).astype(np.uint8), cons)
            r = np.where((x >> np.uint32(18)) < t_gap, np.uint8(4), r)
            acc += np.left_shift(np.uint64(1), r.astype(np.uint64) << np.uint64(3))
        return acc.view(np.uint8).reshape(len(cons), 8)

    return DpInputs(counts(cons_a, rows_a), off_a, counts(cons_b, rows_b), off_b)


def ragged_lengths(seed: int, n_pairs: int, median: float = 1500.0, sigma: float = 0.6, lo: int = 200, hi: int = 8000,
                   jitter: float = 0.05):
    """Segment lengths for the stand-in of BASELINE.json configs[2] (nucmer is not in the image, so there are no real
    anchors): la log-normal (median, sigma) clipped to [lo, hi]; lb = la * (1 + N(0, jitter)), clipped likewise."""
    rng = np.random.default_rng(seed)
    la = np.clip(np.exp(rng.normal(np.log(median), sigma, size=n_pairs)), lo, hi).astype(np.int64)
    lb = np.clip(la * (1.0 + rng.normal(0.0, jitter, size=n_pairs)), lo, hi).astype(np.int64)
    return la, lb
