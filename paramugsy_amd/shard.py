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


def partition_weighted(weights: Sequence[int], world: int) -> List[int]:
    """Contiguous slices of items of unequal cost, cut where the running sum of the weights comes closest to k / world of the total:
    world + 1 positions (cuts[0] = 0, cuts[world] = len(weights)); rank r holds [cuts[r], cuts[r + 1]).  Equal weights: partition()'s
    slices.  The rule of csrc/multi.hpp::partition_weighted (pm_partition_weighted), restated for the ranks of a torch.distributed
    job: a ragged batch cut by count leaves slices of very different cell counts."""
    n = len(weights)
    w = [max(int(x), 0) for x in weights]
    total = sum(w)
    if n == 0 or total <= 0 or all(x == w[0] for x in w):
        return [partition(n, world, r)[0] for r in range(world)] + [n]
    cuts, run, at = [0], 0, 0
    for k in range(1, world):
        # exact rational comparison (the C side compares the same cross products in unsigned __int128, csrc/multi.hpp; the tests compare the two on random inputs)
        while at < n and (run + w[at]) * world - total * k <= total * k - run * world:
            run += w[at]
            at += 1
        cuts.append(at)
    cuts.append(n)
    return cuts


def pair_weights(la, lb):
    """The weight of a profile pair in the DP's partition: its cells, plus its columns (empty profiles still count)."""
    import numpy as np
    la = np.asarray(la, dtype=np.int64)
    lb = np.asarray(lb, dtype=np.int64)
    return la * lb + la + lb + 1


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


def _comm_device(dist):
    """Tensors of the host-side gather live where the process group can move them: the GPU for RCCL, host memory for gloo."""
    import torch
    try:
        if dist.get_backend() == "nccl":
            return torch.device("cuda", torch.cuda.current_device())
    except Exception:
        pass
    return torch.device("cpu")


def _all_ok(ok: bool, rank: int, world: int, dist, what: str, err: Optional[BaseException]) -> None:
    """Every rank learns whether every rank's compute succeeded; if one failed, all raise (none is left waiting in a gather)."""
    import torch
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=_comm_device(dist))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if err is not None:
            raise RuntimeError("%s failed on rank %d: %s" % (what, rank, err)) from err
        raise RuntimeError("%s failed on another rank" % what)


def gather_bytes(mine: bytes, rank: int, world: int, dist) -> Optional[List[bytes]]:
    """Rank-ordered list of every rank's byte string on rank 0 (None elsewhere), as flat uint8 tensors: sizes first, then one
    point-to-point transfer per rank; nothing is pickled."""
    import numpy as np
    import torch
    dev = _comm_device(dist)
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    sizes[rank] = len(mine)
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM)
    sizes = [int(x) for x in sizes.tolist()]
    if rank != 0:
        if len(mine) > 0:
            dist.send(torch.from_numpy(np.frombuffer(mine, dtype=np.uint8).copy()).to(dev), dst=0)
        return None
    parts = [mine]
    for r in range(1, world):
        buf = torch.empty(sizes[r], dtype=torch.uint8, device=dev)
        if sizes[r] > 0:
            dist.recv(buf, src=r)
        parts.append(buf.cpu().numpy().tobytes())
    return parts


def translate_sharded(left_dir: str, right_dir: str, delta_paths: Sequence[str], out_path: str, rank: int, world: int,
                      dist=None, device: int = 0,
                      translate_fn: Optional[Callable[[str, str, Sequence[str], str], None]] = None) -> None:
    """Every rank translates its slice of the delta-file list; rank 0 writes the merged output.

    `dist` is torch.distributed (already initialised; RCCL on GPUs, gloo in the CPU tests) and is used only for
    the host-side gather.  `translate_fn(left_dir, right_dir, paths, out_path)` defaults to the GPU path.  If the
    translation fails on any rank, every rank raises."""
    if translate_fn is None:
        from .translate import translate

        def translate_fn(l, r, paths, out):  # noqa: E306
            translate(l, r, paths, out, device=device)
    lo, hi = partition(len(delta_paths), world, rank)
    fd, tmp = tempfile.mkstemp(prefix="pm_shard_%d_" % rank, suffix=".delta")
    os.close(fd)
    mine, err = b"", None
    try:
        translate_fn(left_dir, right_dir, list(delta_paths[lo:hi]), tmp)
        with open(tmp, "rb") as f:
            mine = f.read()
    except Exception as e:  # noqa: BLE001 -- reported to every rank below
        err = e
    finally:
        os.unlink(tmp)
    if world == 1 or dist is None:
        if err is not None:
            raise err
        parts = [mine]
    else:
        _all_ok(err is None, rank, world, dist, "translate", err)
        parts = gather_bytes(mine, rank, world, dist)
    if rank == 0:
        # the file header names the directories, which are the same for every shard; the merged file is
        # what one process over the whole list prints
        with open(out_path, "wb") as f:
            f.write(merge_delta_outputs(parts))


def gather_pair_results(scores, paths, rank: int, world: int, dist=None):
    """DP: rank-ordered concatenation of per-shard scores and paths on rank 0 (None elsewhere).  Three flat arrays travel per
    rank (scores int32, path lengths int32, all ops uint8 back to back); the per-pair views are rebuilt on rank 0."""
    if world == 1 or dist is None:
        return scores, paths
    import numpy as np
    scores = np.ascontiguousarray(scores, dtype=np.int32)
    n_ops = np.array([len(p) for p in paths], dtype=np.int32)
    flat = np.concatenate([np.asarray(p, dtype=np.uint8) for p in paths]) if len(paths) else np.zeros(0, dtype=np.uint8)
    g_scores = gather_bytes(scores.tobytes(), rank, world, dist)
    g_nops = gather_bytes(n_ops.tobytes(), rank, world, dist)
    g_ops = gather_bytes(flat.tobytes(), rank, world, dist)
    if rank != 0:
        return None, None
    all_scores = np.concatenate([np.frombuffer(b, dtype=np.int32) for b in g_scores])
    all_paths = []
    for nb, ob in zip(g_nops, g_ops):
        lens = np.frombuffer(nb, dtype=np.int32)
        ops = np.frombuffer(ob, dtype=np.uint8)
        ends = np.cumsum(lens)
        all_paths += [ops[e - l:e] for l, e in zip(lens, ends)]
    return all_scores, all_paths


def slice_pairs(inputs, lo: int, hi: int):
    """Pairs [lo, hi) of a DpInputs as a DpInputs of their own (offsets rebased to 0)."""
    import numpy as np
    from .dp import DpInputs
    a0, a1 = int(inputs.off_a[lo]), int(inputs.off_a[hi])
    b0, b1 = int(inputs.off_b[lo]), int(inputs.off_b[hi])
    return DpInputs(inputs.cols_a[a0:a1], (np.asarray(inputs.off_a[lo:hi + 1]) - a0).astype(np.int64), inputs.cols_b[b0:b1],
                    (np.asarray(inputs.off_b[lo:hi + 1]) - b0).astype(np.int64))


def align_sharded(inputs, params, rank: int, world: int, dist=None, device: int = 0, align_fn=None):
    """DP over a static pair partition: rank r aligns its contiguous slice of the pairs -- cut by cells, partition_weighted over
    pair_weights, so that the slices of a ragged batch hold equal work -- on its GPU, rank 0 receives every shard's scores and
    paths in rank order (host-side gather, no data-path collective).
    align_fn(sub_inputs, params) -> (scores, paths) replaces the HIP path in the CPU tests (the oracle).  If the alignment
    fails on any rank, every rank raises."""
    import numpy as np
    cuts = partition_weighted(pair_weights(np.diff(inputs.off_a), np.diff(inputs.off_b)), world)
    lo, hi = cuts[rank], cuts[rank + 1]
    sub = slice_pairs(inputs, lo, hi)
    scores, paths, err = None, None, None
    try:
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
    except Exception as e:  # noqa: BLE001 -- reported to every rank below
        err = e
    if world == 1 or dist is None:
        if err is not None:
            raise err
        return scores, paths
    _all_ok(err is None, rank, world, dist, "align", err)
    return gather_pair_results(scores, paths, rank, world, dist)


def align_blocks_sharded(blocks_a, blocks_b, params, rank: int, world: int, dist=None, device: int = 0, block_fn=None):
    """MAF blocks in, merged MAF blocks out, over a static pair partition: pair k = block k of each side.  Rank r packs, aligns
    and expands its contiguous slice on its GPU (pm_dp_align_blocks: pack -> DP -> expansion without leaving the device); rank 0 receives every
    shard's scores and merged blocks in pair order: the "host-side gather of MAF blocks".  Three flat arrays travel per rank
    (scores, rows and columns per merged block, all row texts back to back).
    block_fn(sub_a, sub_b, params) -> (scores, merged blocks) replaces the HIP path in the CPU tests (the oracle)."""
    import numpy as np
    cuts = partition_weighted(pair_weights([len(b[0]) if b else 0 for b in blocks_a], [len(b[0]) if b else 0 for b in blocks_b]), world)
    lo, hi = cuts[rank], cuts[rank + 1]
    sub_a, sub_b = blocks_a[lo:hi], blocks_b[lo:hi]
    scores, merged, err = None, None, None
    try:
        if block_fn is not None:
            scores, merged = block_fn(sub_a, sub_b, params)
        else:
            from . import dp
            scores, merged = dp.align_blocks(sub_a, sub_b, params, device=device)
    except Exception as e:  # noqa: BLE001 -- reported to every rank below
        err = e
    if world == 1 or dist is None:
        if err is not None:
            raise err
        return np.asarray(scores, dtype=np.int32), merged
    _all_ok(err is None, rank, world, dist, "align_blocks", err)
    shape = np.array([[len(m), len(m[0]) if m else 0] for m in merged], dtype=np.int64).reshape(-1, 2)
    g_scores = gather_bytes(np.ascontiguousarray(scores, dtype=np.int32).tobytes(), rank, world, dist)
    g_shape = gather_bytes(shape.tobytes(), rank, world, dist)
    g_text = gather_bytes(b"".join(r for m in merged for r in m), rank, world, dist)
    if rank != 0:
        return None, None
    all_scores = np.concatenate([np.frombuffer(b, dtype=np.int32) for b in g_scores])
    all_blocks = []
    for sb, tb in zip(g_shape, g_text):
        at = 0
        for rows, cols in np.frombuffer(sb, dtype=np.int64).reshape(-1, 2):
            all_blocks.append([tb[at + r * cols: at + (r + 1) * cols] for r in range(int(rows))])
            at += int(rows) * int(cols)
    return all_scores, all_blocks
