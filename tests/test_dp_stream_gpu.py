"""pm_dp_stream_*: a batch uploaded in segments with the fill kernel running behind the uploads must give exactly what the
resident batch (pm_dp_batch_*) gives -- which the other tests pin to the scalar oracle -- whatever the number of segments."""
import numpy as np
import pytest

from paramugsy_amd import dp

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def segments_as_asked(monkeypatch):
    """The engine uses fewer segments than asked for when they would hold less than 5e9 cells each; these tests want tiny batches in
    many segments."""
    monkeypatch.setenv("PM_DP_SEGMENT_CELLS", "1")


def resident(inputs, params):
    b = dp.DpBatch(inputs, params)
    b.run(True)
    scores, ops, n_ops = b.fetch()
    b.close()
    return scores, ops, n_ops


@pytest.mark.parametrize("segments", [1, 2, 5, 17, 1000])
def test_stream_equals_resident_batch_on_ragged_pairs(segments, oracle_build, monkeypatch):
    import pyoracle
    monkeypatch.setenv("PM_DP_MODE", "ckpt" if segments % 2 else "bits")  # a batch this small would always choose bits
    la, lb = dp.ragged_lengths(5, 60, median=400, sigma=0.7, lo=1, hi=2500)
    la[3], lb[3] = 0, 17
    la[4], lb[4] = 9, 0
    inputs = dp.synth_batch(6, la, lb, 3, 3)
    params = dp.make_params(3, 3)
    r_scores, r_ops, r_nops = resident(inputs, params)
    st = dp.DpStream(params, segments)
    for _ in range(2):  # the slots are reused: a second pass over the same engine
        scores, ops, n_ops = st.align(inputs)
        assert np.array_equal(scores, r_scores) and np.array_equal(n_ops, r_nops)
        for p, q in zip(dp.paths_of(inputs, ops, n_ops), dp.paths_of(inputs, r_ops, r_nops)):
            assert np.array_equal(p, q)
    s_only, _, _ = st.align(inputs, with_paths=False)
    assert np.array_equal(s_only, r_scores)
    st.close()
    o_scores, o_paths = pyoracle.dp_align(inputs, params)
    assert np.array_equal(r_scores, o_scores)


@pytest.mark.parametrize("tiers", [False, True])
def test_small_batch_in_two_segments_runs_as_one_chunk_in_the_input_order(tiers, oracle_build, monkeypatch):
    """A batch of fewer than four segments' worth of cells stays ONE chunk in the caller's order with a fill launch per segment (the
    other tests here, with segments of one cell, get a chunk per segment, its pairs longest first, the long ones in tiers)."""
    monkeypatch.setenv("PM_DP_MODE", "ckpt")
    la, lb = dp.ragged_lengths(9, 90, median=300, sigma=0.8, lo=1, hi=3000)
    inputs = dp.synth_batch(10, la, lb, 2, 2)
    params = dp.make_params(2, 2)
    cells = int(np.sum(la.astype(np.int64) * lb))
    monkeypatch.setenv("PM_DP_SEGMENT_CELLS", str(cells // 3 + 1))
    if tiers:
        monkeypatch.setenv("PM_DP_TIER_MIN_PAIRS", "8")
    r_scores, r_ops, r_nops = resident(inputs, params)
    for segments in (2, 6):
        st = dp.DpStream(params, segments)
        for _ in range(2):
            scores, ops, n_ops = st.align(inputs)
            assert np.array_equal(scores, r_scores) and np.array_equal(n_ops, r_nops)
            for p, q in zip(dp.paths_of(inputs, ops, n_ops), dp.paths_of(inputs, r_ops, r_nops)):
                assert np.array_equal(p, q)
        st.close()


@pytest.mark.parametrize("segments", [1, 3, 8])
def test_ragged_batch_with_tiers_in_a_chunk_per_segment(segments, oracle_build, monkeypatch):
    """Chunks that end where segments end, ordered longest first, the longest pairs of every chunk in tiers of their own (the
    mark for tiers lowered to 8 pairs), results leaving chunk by chunk into pinned arrays: equal to the resident batch."""
    monkeypatch.setenv("PM_DP_MODE", "ckpt")
    monkeypatch.setenv("PM_DP_TIER_MIN_PAIRS", "8")
    la, lb = dp.ragged_lengths(21, 400, median=250, sigma=0.9, lo=1, hi=4000)
    src = dp.synth_batch(22, la, lb, 4, 4)
    params = dp.make_params(4, 4)
    r_scores, r_ops, r_nops = resident(src, params)
    pa, pb = dp.PinnedArray(src.cols_a.shape, np.uint8), dp.PinnedArray(src.cols_b.shape, np.uint8)
    pa.a[...] = src.cols_a
    pb.a[...] = src.cols_b
    inputs = dp.DpInputs(pa.a, src.off_a, pb.a, src.off_b)
    n = len(la)
    ps, pn = dp.PinnedArray((n,), np.int32), dp.PinnedArray((n,), np.int32)
    po = dp.PinnedArray((max(1, int(src.off_a[-1] + src.off_b[-1])),), np.uint8)
    st = dp.DpStream(params, segments)
    for _ in range(2):
        scores, ops, n_ops = st.align(inputs, ps.a, po.a, pn.a)
        assert np.array_equal(scores, r_scores) and np.array_equal(n_ops, r_nops)
        for p, q in zip(dp.paths_of(inputs, ops, n_ops), dp.paths_of(inputs, r_ops, r_nops)):
            assert np.array_equal(p, q)
    st.close()
    for x in (pa, pb, ps, pn, po):
        x.close()


def _used_ops_equal(inputs, ops_x, ops_y, n_ops):
    """The ops of every pair (the right-aligned, used part of its slot) equal in the two arrays, without a Python loop."""
    end = (inputs.off_a[1:] + inputs.off_b[1:]).astype(np.int64)
    start = end - n_ops.astype(np.int64)
    mark = np.zeros(int(end[-1]) + 1, dtype=np.int32)
    np.add.at(mark, start, 1)
    np.add.at(mark, end, -1)
    used = np.cumsum(mark[:-1]) > 0
    return bool(np.array_equal(ops_x[:len(used)][used], ops_y[:len(used)][used]))


@pytest.mark.parametrize("shape", ["uniform", "ragged"])
def test_large_batches_take_the_engines_own_choices(shape, monkeypatch):
    """Batches of more than 1e11 cells with the engine's defaults (no segment size forced): a short first segment, a chunk per
    segment ordered longest first inside, tiers for the ragged one, results leaving chunk by chunk -- every score, path length and
    op equal to the resident batch's (which tests/test_dp_full_gpu.py holds to the oracle at these sizes)."""
    import torch
    from paramugsy_amd.synth_device import synth_batch_device
    monkeypatch.delenv("PM_DP_SEGMENT_CELLS", raising=False)
    if shape == "uniform":
        n, rows = 9000, 2
        la = lb = np.full(n, 4096, dtype=np.int64)
    else:
        n, rows = 52000, 4
        la, lb = dp.ragged_lengths(31, n)
    inputs = synth_batch_device(77, la, lb, rows, rows)
    assert float(np.sum(la.astype(np.float64) * lb)) > 1e11
    params = dp.make_params(rows, rows)
    r_scores, r_ops, r_nops = resident(inputs, params)
    pa, pb = dp.PinnedArray(inputs.cols_a.shape, np.uint8), dp.PinnedArray(inputs.cols_b.shape, np.uint8)
    pa.a[...] = inputs.cols_a
    pb.a[...] = inputs.cols_b
    pin = dp.DpInputs(pa.a, inputs.off_a, pb.a, inputs.off_b)
    ps, pn = dp.PinnedArray((n,), np.int32), dp.PinnedArray((n,), np.int32)
    po = dp.PinnedArray((int(inputs.off_a[-1] + inputs.off_b[-1]),), np.uint8)
    st = dp.DpStream(params, 8)
    for _ in range(2):
        ps.a[...] = 0
        pn.a[...] = 0
        st.align(pin, ps.a, po.a, pn.a)
        assert np.array_equal(ps.a, r_scores) and np.array_equal(pn.a, r_nops)
        assert _used_ops_equal(inputs, po.a, r_ops, r_nops)
    st.close()
    for x in (pa, pb, ps, pn, po):
        x.close()
    torch.cuda.empty_cache()


def test_stream_with_pinned_buffers_and_several_workspace_chunks(oracle_build, monkeypatch):
    monkeypatch.setenv("PM_DP_MODE", "ckpt")
    """Pinned inputs and outputs (the asynchronous case) and a workspace so small that the batch takes several chunks, each of
    them spanning several upload segments."""
    n, rows, L = 600, 2, 500
    src = dp.synth_pairs_fast(11, n, rows, L)
    params = dp.make_params(rows, rows)
    r_scores, r_ops, r_nops = resident(src, params)
    pa, pb = dp.PinnedArray(src.cols_a.shape, np.uint8), dp.PinnedArray(src.cols_b.shape, np.uint8)
    pa.a[...] = src.cols_a
    pb.a[...] = src.cols_b
    inputs = dp.DpInputs(pa.a, src.off_a, pb.a, src.off_b)
    ps, po, pn = dp.PinnedArray((n,), np.int32), dp.PinnedArray((2 * n * L,), np.uint8), dp.PinnedArray((n,), np.int32)
    st = dp.DpStream(params, 7, workspace_bytes=60 << 20)
    scores, ops, n_ops = st.align(inputs, ps.a, po.a, pn.a)
    assert np.array_equal(scores, r_scores) and np.array_equal(n_ops, r_nops)
    for p, q in zip(dp.paths_of(inputs, ops, n_ops), dp.paths_of(inputs, r_ops, r_nops)):  # the slots' unused heads are not compared
        assert np.array_equal(p, q)
    st.close()
    for x in (pa, pb, ps, po, pn):
        x.close()


def test_stream_refuses_what_the_batch_refuses():
    from paramugsy_amd import capi
    cols = np.zeros((4, 8), dtype=np.uint8)
    cols[:, :5] = 255
    off = np.array([0, 4], dtype=np.int64)
    p = dp.make_params(1, 1, match=127, mismatch=-127)
    st = dp.DpStream(p, 2)
    with pytest.raises(capi.PmError) as e:
        st.align(dp.DpInputs(cols, off, cols.copy(), off.copy()))
    assert e.value.code == capi.PM_E_INVALID
    st.close()


def test_variant_chosen_from_the_first_segment_is_corrected(oracle_build):
    """The first segment's counts fit the int8 kernel, a later segment's do not: the engine must notice once the whole batch is up
    and run again with int16 weights -- results equal the resident batch's."""
    rng = np.random.default_rng(9)
    n, L = 40, 120

    def cols(rows):
        c = np.zeros((n * L, 8), dtype=np.uint8)
        pick = rng.integers(0, 5, size=(n * L, 3))
        for s_ in range(5):
            c[:, s_] = (pick == s_).sum(axis=1)
        return c
    ca, cb = cols(3), cols(3)
    cb[(n - 3) * L:, 0] = 90  # deep columns at the very end of B: 90 x max|sub| > 127
    off = np.arange(n + 1, dtype=np.int64) * L
    inputs = dp.DpInputs(ca, off, cb, off.copy())
    params = dp.make_params(1, 1, open_per_pair=60, extend_per_pair=5)
    r_scores, r_ops, r_nops = resident(inputs, params)
    b = dp.DpBatch(inputs, params)
    assert not b.variant()["dot4"]
    b.close()
    st = dp.DpStream(params, 8)
    scores, ops, n_ops = st.align(inputs)
    st.close()
    assert np.array_equal(scores, r_scores) and np.array_equal(n_ops, r_nops)
    for p, q in zip(dp.paths_of(inputs, ops, n_ops), dp.paths_of(inputs, r_ops, r_nops)):
        assert np.array_equal(p, q)


@pytest.mark.parametrize("segments", [1, 4, 9])
def test_row_texts_in_equal_packed_columns_in(segments, oracle_build, monkeypatch):
    """pm_dp_stream_align_text: the rows of MAF blocks go up and are packed on the device segment by segment; scores and paths
    equal the resident batch over the columns pm_dp_pack_maf makes of the same rows -- blocks of different depths and widths,
    lower case, N's, an empty block on either side."""
    monkeypatch.setenv("PM_DP_MODE", "ckpt" if segments % 2 else "bits")
    from test_dp_maf import random_blocks
    rng = np.random.default_rng(segments)
    A = random_blocks(rng, 37, max_rows=5, max_cols=700)
    B = random_blocks(rng, 37, max_rows=5, max_cols=700)
    A[5], B[9] = [], []
    params = dp.make_params(3, 3)
    ca, oa = dp.pack_maf(A)
    cb, ob = dp.pack_maf(B)
    inputs = dp.DpInputs(ca, oa, cb, ob)
    r_scores, r_ops, r_nops = resident(inputs, params)
    st = dp.DpStream(params, segments)
    for _ in range(2):
        scores, ops, n_ops = st.align_text(dp.flatten_blocks(A), dp.flatten_blocks(B))
        assert np.array_equal(scores[:37], r_scores) and np.array_equal(n_ops[:37], r_nops)
        for p, q in zip(dp.paths_of(inputs, ops, n_ops[:37]), dp.paths_of(inputs, r_ops, r_nops)):
            assert np.array_equal(p, q)
    s_only, _, _ = st.align_text(dp.flatten_blocks(A), dp.flatten_blocks(B), with_paths=False)
    assert np.array_equal(s_only[:37], r_scores)
    # the column engine still works on the same stream object afterwards
    scores, ops, n_ops = st.align(inputs)
    assert np.array_equal(scores, r_scores)
    st.close()


def test_pinned_array_outlives_its_wrapper(oracle_build):
    """dp.PinnedArray's `.a` owns the allocation: dropping the wrapper (round 3: that freed the memory under a copy engine that was
    still reading it -- a GPU memory access fault) leaves the array, and every view of it, usable for the upload."""
    import gc
    import pyoracle
    rows, n, L = 3, 40, 700
    src = dp.synth_pairs_fast(77, n, rows, L)
    params = dp.make_params(rows, rows)

    def pinned_copy(x):
        t = dp.PinnedArray(x.shape, x.dtype)
        t.a[...] = x
        return t.a  # the wrapper dies here

    ca, cb = pinned_copy(src.cols_a), pinned_copy(src.cols_b)
    gc.collect()
    junk = [dp.PinnedArray((1 << 20,), np.uint8) for _ in range(4)]  # would reuse freed pinned memory
    for j in junk:
        j.a[...] = 0xff
    inputs = dp.DpInputs(ca, src.off_a, cb[:], src.off_b)
    st = dp.DpStream(params, 2)
    scores, ops, n_ops = st.align(inputs)
    st.close()
    o_scores, o_paths = pyoracle.dp_align(src, params)
    assert np.array_equal(scores, o_scores)
    assert all(np.array_equal(p, q) for p, q in zip(dp.paths_of(src, ops, n_ops), o_paths))


def test_row_texts_from_pinned_memory_large_batch(oracle_build):
    """BASELINE configs[1]'s shape through the text entry from pinned buffers: 2-row x 1 kbp pairs, 2 bytes per column up instead
    of 8; a sample under the oracle."""
    import pyoracle
    from paramugsy_amd.shard import slice_pairs
    n, rows, L = 3000, 2, 1000
    inputs, side_a, side_b = dp.synth_pairs_fast(23, n, rows, L, with_rows=True)
    params = dp.make_params(rows, rows)
    pins = []

    def pinned(side):
        t = dp.PinnedArray(side[0].shape, np.uint8)
        t.a[...] = side[0]
        pins.append(t)
        return (t.a, side[1], side[2])
    st = dp.DpStream(params, 4)
    scores, ops, n_ops = st.align_text(pinned(side_a), pinned(side_b))
    st.close()
    r_scores, r_ops, r_nops = resident(inputs, params)
    assert np.array_equal(scores, r_scores) and np.array_equal(n_ops, r_nops)
    o_scores, o_paths = pyoracle.dp_align(slice_pairs(inputs, 0, 25), params)
    assert np.array_equal(scores[:25], o_scores)
    assert all(np.array_equal(p, q) for p, q in zip(dp.paths_of(inputs, ops, n_ops)[:25], o_paths))
    for t in pins:
        t.close()
