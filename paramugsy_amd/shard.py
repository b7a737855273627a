"""Static sharding of a job over the GPUs of one node, and the host-side gather of its output.

The path shards with no exchange step: work units are independent and every shard reads the
same two read-only sides, so ranks never talk on the data path -- one process per GPU, a contiguous slice of
the delta-file list (translate) or of the pair list (DP) each, and a gather of the outputs on rank 0.  The
reference's own analogue is one OS process per job (lib/base/queued_task_server.ml:57-66).

The gather must be order-preserving: M_delta_stream_writer prints a `>` header only when the name pair
differs from the previous entry's (lib/profiles_lib/m_delta_stream_writer.hh:62-67) and keeps that state across
delta files, so a shard's leading header is dropped when the previous shard ended under the same header.
"""
from __future__ import annotations

import os
import tempfile
from typing import Callable, List, Optional, Sequence, Tuple


def partition(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced: the first n_items % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def split_header(delta_text: bytes) -> Tuple[bytes, bytes]:
    """(the two file header lines of m_translate_main.cc:35-39, the rest)."""
    a = delta_text.find(b"\n")
    b = delta_text.find(b"\n", a + 1)
    return delta_text[:b + 1], delta_text[b + 1:]


def _last_header(body: bytes) -> Optional[bytes]:
    at = body.rfind(b"\n>")
    if at >= 0:
        end = body.find(b"\n", at + 1)
        return body[at + 1:end + 1]
    if body.startswith(b">"):
        return body[:body.find(b"\n") + 1]
    return None


def merge_delta_outputs(parts: Sequence[bytes]) -> bytes:
    """Rank-ordered outputs of `m_translate` runs over consecutive slices of one delta-file list -> the bytes a
    single run over the whole list prints."""
    if not parts:
        return b""
    header, _ = split_header(parts[0])
    out: List[bytes] = [header]
    in_force: Optional[bytes] = None
    for p in parts:
        _, body = split_header(p)
        if not body:
            continue
        if in_force is not None and body.startswith(in_force):
            body = body[len(in_force):]
        out.append(body)
        last = _last_header(body)
        if last is not None:
            in_force = last
    return b"".join(out)


def translate_sharded(left_dir: str, right_dir: str, delta_paths: Sequence[str], out_path: str, rank: int, world: int,
                      dist=None, device: int = 0,
                      translate_fn: Optional[Callable[[str, str, Sequence[str], str], None]] = None) -> None:
    """Every rank translates its slice of the delta-file list; rank 0 writes the merged output.

    `dist` is torch.distributed (already initialised; RCCL on GPUs, gloo in the CPU tests) and is used only for
    the host-side gather.  `translate_fn(left_dir, right_dir, paths, out_path)` defaults to the GPU path."""
    if translate_fn is None:
        from .translate import translate

        def translate_fn(l, r, paths, out):  # noqa: E306
            translate(l, r, paths, out, device=device)
    lo, hi = partition(len(delta_paths), world, rank)
    fd, tmp = tempfile.mkstemp(prefix="pm_shard_%d_" % rank, suffix=".delta")
    os.close(fd)
    try:
        translate_fn(left_dir, right_dir, list(delta_paths[lo:hi]), tmp)
        with open(tmp, "rb") as f:
            mine = f.read()
    finally:
        os.unlink(tmp)
    if world == 1 or dist is None:
        parts = [mine]
    else:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        parts = gathered
    if rank == 0:
        # the file header names the directories, which are the same for every shard; the merged file is
        # what one process over the whole list prints
        with open(out_path, "wb") as f:
            f.write(merge_delta_outputs(parts))


def gather_pair_results(scores, paths, rank: int, world: int, dist=None):
    """DP: rank-ordered concatenation of per-shard scores and paths on rank 0 (None elsewhere)."""
    if world == 1 or dist is None:
        return scores, paths
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((scores, paths), gathered, dst=0)
    if rank != 0:
        return None, None
    import numpy as np
    return np.concatenate([g[0] for g in gathered]), [p for g in gathered for p in g[1]]


def slice_pairs(inputs, lo: int, hi: int):
    """Pairs [lo, hi) of a DpInputs as a DpInputs of their own (offsets rebased to 0)."""
    import numpy as np
    from .dp import DpInputs
    a0, a1 = int(inputs.off_a[lo]), int(inputs.off_a[hi])
    b0, b1 = int(inputs.off_b[lo]), int(inputs.off_b[hi])
    return DpInputs(inputs.cols_a[a0:a1], (np.asarray(inputs.off_a[lo:hi + 1]) - a0).astype(np.int64), inputs.cols_b[b0:b1],
                    (np.asarray(inputs.off_b[lo:hi + 1]) - b0).astype(np.int64))


def align_sharded(inputs, params, rank: int, world: int, dist=None, device: int = 0, align_fn=None):
    """DP over a static pair partition: rank r aligns the contiguous slice partition(n_pairs, world, r) on its GPU,
    rank 0 receives every shard's scores and paths in rank order (host-side gather, no data-path collective).
    align_fn(sub_inputs, params) -> (scores, paths) replaces the HIP path in the CPU tests (the oracle)."""
    lo, hi = partition(inputs.n_pairs, world, rank)
    sub = slice_pairs(inputs, lo, hi)
    if align_fn is not None:
        scores, paths = align_fn(sub, params)
    else:
        from .dp import DpBatch
        batch = DpBatch(sub, params, device=device)
        try:
            batch.run(traceback=True)
            scores, ops, n_ops = batch.fetch()
            paths = batch.paths(ops, n_ops)
        finally:
            batch.close()
    return gather_pair_results(scores, paths, rank, world, dist)
