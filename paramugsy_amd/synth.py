"""Seeded synthetic inputs for the "profiles" translate path.

There is no network and no MUMmer/Mugsy in the build image, so every workload is synthetic.  The
generator works at the level of the reference's on-disk formats, so that the same bytes can be fed to
the reference binary, the oracle and the HIP path:

* a *side* (left or right) is a multi-genome alignment: blocks of rows, each row a gapped text over one
  genome interval (what a MAF block is).  ``rows_to_profiles_text`` renders it in the ``profiles`` record
  layout written by lib/profiles/m_profile.ml:122-135 and read by lib/profiles_lib/m_profile.cc:15-85;
  ``side_to_maf_text`` renders the same side as MAF (input of ``mugsy_profiles make``).
* a *delta file* is MUMmer's .delta text as lib/profiles_lib/m_delta.cc:72-92,148-220 parses it.

Nothing here is on the product's hot path; bench.py and the tests call it to make inputs.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


@dataclass
class Row:
    """One `s` line of a MAF block == one row profile."""
    seq_name: str
    fwd_start: int      # 1-based first base on the forward strand
    size: int           # bases (non-gap columns)
    forward: bool
    src_size: int
    text: bytes         # gapped text, len == block columns

    @property
    def maf_start(self) -> int:
        # inverse of of_maf, lib/profiles_lib/m_range.hh:106-115
        if self.forward:
            return self.fwd_start - 1
        return self.src_size - (self.fwd_start + self.size - 1)

    @property
    def prange(self) -> Tuple[int, int]:
        if self.forward:
            return (self.fwd_start, self.fwd_start + self.size - 1)
        return (self.fwd_start + self.size - 1, self.fwd_start)


@dataclass
class Block:
    rows: List[Row] = field(default_factory=list)


def gaps_of_text(text: bytes) -> List[Tuple[int, int]]:
    """1-based inclusive runs of '-' (lib/profiles/m_profile.ml:29-47)."""
    a = np.frombuffer(text, dtype=np.uint8) == ord("-")
    if not a.any():
        return []
    d = np.diff(np.concatenate(([0], a.astype(np.int8), [0])))
    starts = np.nonzero(d == 1)[0] + 1
    ends = np.nonzero(d == -1)[0]
    return list(zip(starts.tolist(), ends.tolist()))


def _gapped_text(rng: np.random.Generator, columns: int, gap_rate: float, mean_gap: float,
                 edge_gap_prob: float) -> bytes:
    """Random row text over `columns` columns with geometric gap runs; at least one base."""
    is_gap = np.zeros(columns, dtype=bool)
    n_runs = rng.poisson(columns * gap_rate)
    for _ in range(int(n_runs)):
        at = int(rng.integers(0, columns))
        ln = int(rng.geometric(1.0 / mean_gap))
        is_gap[at:at + ln] = True
    if rng.random() < edge_gap_prob:
        is_gap[:int(rng.geometric(1.0 / mean_gap))] = True
    if rng.random() < edge_gap_prob:
        is_gap[columns - int(rng.geometric(1.0 / mean_gap)):] = True
    if is_gap.all():
        is_gap[int(rng.integers(0, columns))] = False
    txt = BASES[rng.integers(0, 4, size=columns)].copy()
    txt[is_gap] = ord("-")
    return txt.tobytes()


def gen_side(rng: np.random.Generator, genomes: Sequence[str], genome_len: int, n_blocks: int,
             mean_cols: int = 400, gap_rate: float = 0.01, mean_gap: float = 3.0,
             row_prob: float = 0.8, rev_prob: float = 0.2, edge_gap_prob: float = 0.15,
             spacing: int = 40, overlap_prob: float = 0.0) -> List[Block]:
    """A side: `n_blocks` blocks; per genome the covered intervals are disjoint and ascending -- unless overlap_prob > 0: then a
    row starts, with that probability, somewhere inside the stretch the genome's earlier rows cover (rows that overlap, nest and
    come out of file order once sorted by start: what the per-sequence row index and its binary search have to cope with)."""
    cursor = {g: 1 + int(rng.integers(0, spacing)) for g in genomes}
    blocks: List[Block] = []
    for _ in range(n_blocks):
        columns = max(4, int(rng.normal(mean_cols, mean_cols / 4)))
        blk = Block()
        for g in genomes:
            if rng.random() > row_prob:
                continue
            text = _gapped_text(rng, columns, gap_rate, mean_gap, edge_gap_prob)
            size = columns - text.count(b"-")
            start = cursor[g] + int(rng.integers(0, spacing))
            if overlap_prob > 0 and rng.random() < overlap_prob:
                start = int(rng.integers(1, max(2, cursor[g])))
            if start + size - 1 > genome_len:
                continue
            cursor[g] = max(cursor[g], start + size)
            blk.rows.append(Row(g, start, size, bool(rng.random() >= rev_prob), genome_len, text))
        if blk.rows:
            blocks.append(blk)
    return blocks


def rows_to_profiles_text(blocks: Sequence[Block], basename: str) -> str:
    """`profiles` file: major name "%s.%s_%04d" (lib/profiles/m_profile_stream.ml:65), minor = row index."""
    out: List[str] = []
    for bi, blk in enumerate(blocks):
        major = "%s.%s_%04d" % (basename, basename, bi)
        for ri, row in enumerate(blk.rows):
            s, e = row.prange
            out.append("%s %d %s %d %d %d %d\n" % (major, ri, row.seq_name, s, e, len(row.text), row.src_size))
            for gs, ge in gaps_of_text(row.text):
                out.append("%d %d\n" % (gs, ge))
            out.append("0\n")
            out.append(row.text.decode() + "\n")
    return "".join(out)


def side_to_maf_text(blocks: Sequence[Block]) -> str:
    out = ["##maf version=1 scoring=paramugsy\n"]
    for blk in blocks:
        out.append("a score=0 label=1 mult=%d\n" % len(blk.rows))
        for row in blk.rows:
            out.append("s %s %d %d %s %d %s\n" % (row.seq_name, row.maf_start, row.size,
                                                 "+" if row.forward else "-", row.src_size, row.text.decode()))
        out.append("\n")
    return "".join(out)


def write_side(dir_path: str, blocks: Sequence[Block], basename: str) -> None:
    os.makedirs(dir_path, exist_ok=True)
    with open(os.path.join(dir_path, "profiles"), "w") as f:
        f.write(rows_to_profiles_text(blocks, basename))


def _delta_offsets(rng: np.random.Generator, ref_bases: int, indel_rate: float, mean_indel: float,
                   adjacent_prob: float) -> Tuple[List[int], int]:
    """Signed offsets for an alignment that consumes exactly `ref_bases` reference bases.

    Returns (offsets without the terminating 0, query bases consumed).  Negative = gap in the reference
    row, positive = gap in the query row (lib/profiles_lib/m_delta.cc:14-41).
    """
    offsets: List[int] = []
    ref_left = ref_bases
    qry = 0
    since = 0  # columns since the previous gap's last column
    first = True
    while ref_left > 1:
        run = int(rng.geometric(indel_rate))
        if not first and rng.random() < adjacent_prob:
            run = 0
        run = min(run, ref_left - 1)
        if first:
            run = max(run, 1)
        first = False
        ref_left -= run
        qry += run
        since += run
        ln = int(rng.geometric(1.0 / mean_indel))
        in_query = bool(rng.random() < 0.5)
        if in_query:
            ln = min(ln, ref_left - 1)
            if ln <= 0:
                break
        if offsets and since == 0:
            prev_query = offsets[-1] > 0
            if prev_query == in_query:
                in_query = not in_query  # an adjacent run of the same sign would merge into the previous gap
                if in_query:
                    ln = min(ln, ref_left - 1)
                    if ln <= 0:
                        break
        sign = 1 if in_query else -1
        offsets.append(sign * (since + 1))
        offsets.extend([sign] * (ln - 1))
        if in_query:
            ref_left -= ln   # gap in query row: reference bases with nothing opposite
        else:
            qry += ln        # gap in reference row: query bases with nothing opposite
        since = 0
    qry += ref_left
    return offsets, qry


def gen_delta_text(rng: np.random.Generator, ref_names: Sequence[str], qry_names: Sequence[str],
                   ref_len: int, qry_len: int, n_entries: int, mean_len: int = 1500,
                   indel_rate: float = 0.004, mean_indel: float = 2.0, rev_prob: float = 0.3,
                   adjacent_prob: float = 0.02, group: int = 4) -> str:
    """A .delta file: entries grouped under `>ref qry len len` headers, `group` entries per header."""
    out = ["/synthetic/ref.fasta /synthetic/qry.fasta\n", "NUCMER\n"]
    done = 0
    while done < n_entries:
        rn = ref_names[int(rng.integers(0, len(ref_names)))]
        qn = qry_names[int(rng.integers(0, len(qry_names)))]
        out.append(">%s %s %d %d\n" % (rn, qn, ref_len, qry_len))
        for _ in range(min(group, n_entries - done)):
            ref_bases = max(2, int(rng.normal(mean_len, mean_len / 3)))
            ref_bases = min(ref_bases, ref_len - 1)
            offsets, qry_bases = _delta_offsets(rng, ref_bases, indel_rate, mean_indel, adjacent_prob)
            qry_bases = max(1, qry_bases)
            if qry_bases > qry_len - 1:
                continue
            rs = int(rng.integers(1, ref_len - ref_bases + 2))
            qs = int(rng.integers(1, qry_len - qry_bases + 2))
            re_, qe = rs + ref_bases - 1, qs + qry_bases - 1
            if rng.random() < rev_prob:
                qs, qe = qe, qs
            out.append("%d %d %d %d 0 0 0\n" % (rs, re_, qs, qe))
            for v in offsets:
                out.append("%d\n" % v)
            out.append("0\n")
            done += 1
    return "".join(out)


@dataclass
class Workload:
    """Paths of one generated translate job (the reference CLI's four arguments)."""
    left_dir: str
    right_dir: str
    list_path: str
    delta_paths: List[str]


def make_workload(root: str, seed: int, n_left: int = 3, n_right: int = 3, genome_len: int = 60000,
                  n_blocks: int = 60, n_deltas: int = 2, entries_per_delta: int = 40, **kw) -> Workload:
    """Write a complete m_translate job under `root` and return its paths."""
    rng = np.random.default_rng(seed)
    side_kw = {k: kw[k] for k in ("mean_cols", "gap_rate", "mean_gap", "row_prob", "rev_prob", "edge_gap_prob", "spacing", "overlap_prob") if k in kw}
    delta_kw = {k: kw[k] for k in ("mean_len", "indel_rate", "mean_indel", "adjacent_prob", "group") if k in kw}
    if "delta_rev_prob" in kw:
        delta_kw["rev_prob"] = kw["delta_rev_prob"]
    lg = ["L%d.chr" % k for k in range(n_left)]
    rg = ["R%d.chr" % k for k in range(n_right)]
    os.makedirs(root, exist_ok=True)
    left_dir = os.path.join(root, "profiles-l")
    right_dir = os.path.join(root, "profiles-r")
    write_side(left_dir, gen_side(rng, lg, genome_len, n_blocks, **side_kw), "l")
    write_side(right_dir, gen_side(rng, rg, genome_len, n_blocks, **side_kw), "r")
    paths = []
    for d in range(n_deltas):
        p = os.path.join(root, "nucmer_%d.delta" % d)
        with open(p, "w") as f:
            f.write(gen_delta_text(rng, lg, rg, genome_len, genome_len, entries_per_delta, **delta_kw))
        paths.append(p)
    list_path = os.path.join(root, "nucmer.list")
    with open(list_path, "w") as f:
        f.write("".join(p + "\n" for p in paths))
    return Workload(left_dir, right_dir, list_path, paths)


def shift_positions(tables, left_shift: int, right_shift: int):
    """The same job further along its sequences: every left row and every entry's reference range moved by left_shift bases, every
    right row and query range by right_shift.  What translate writes is in profile COLUMNS, so the result does not change -- which
    makes this both a test property and the way to build a job whose positions need 64 bits (a chromosome beyond 2^25) out of a
    bacterial-sized one.  Returns a new Tables; the input is left alone."""
    import copy
    t = copy.deepcopy(tables)
    for side, shift in ((t.left, left_shift), (t.right, right_shift)):
        side["start"] = side["start"] + np.int64(shift)
        side["end"] = side["end"] + np.int64(shift)
    for k, shift in (("ref_start", left_shift), ("ref_end", left_shift), ("qry_start", right_shift), ("qry_end", right_shift)):
        t.deltas[k] = t.deltas[k] + np.int64(shift)
    return t
