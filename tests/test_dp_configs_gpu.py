"""BASELINE.json's DP configurations at full shape on the GPU, under the oracle (oracle/dp_oracle.c -- this repo's own
specification: the reference has no DP).  The batches are too big for the scalar oracle as a whole, so every
test checks the scores of a SAMPLE of pairs against the oracle's scorer and re-scores the sampled paths under the
specification (a path that spans the pair and re-scores to the optimum is an optimal alignment); the sample always holds the
first and the last pair of several chunks of the traceback workspace."""
import numpy as np
import pytest

from paramugsy_amd import dp
from paramugsy_amd.shard import slice_pairs

pytestmark = pytest.mark.gpu


def check_sample(inputs, params, scores, paths, sample, full_paths_for=()):
    import pyoracle
    sample = sorted(set(int(k) for k in sample))
    for k in sample:
        one = slice_pairs(inputs, k, k + 1)
        assert scores[k] == pyoracle.dp_scores(one, params)[0], "score of pair %d" % k
        rc, s = pyoracle.dp_score_of_path(inputs, params, k, paths[k])
        assert rc == 0 and s == scores[k], "path of pair %d" % k
    for k in full_paths_for:  # the oracle's own path, op for op
        one = slice_pairs(inputs, int(k), int(k) + 1)
        o_scores, o_paths = pyoracle.dp_align(one, params)
        assert o_scores[0] == scores[k] and np.array_equal(o_paths[0], paths[k]), "oracle path of pair %d" % k


@pytest.mark.parametrize("budget_gib", [0, 24])
def test_north_star_shape_one_gpu_share(budget_gib, oracle_build):
    """12 500 pairs of 8 rows x 4 096 columns (one GPU's eighth of the north-star's 100 k): at the default workspace budget (one
    chunk on a 288 GB device) and at 24 GiB, where the chunk pipeline runs (path kernel of chunk c beside fill kernel of c + 1)."""
    n, rows, L = 12500, 8, 4096
    inputs = dp.synth_batch(20261003, np.full(n, L), np.full(n, L), rows, rows)
    params = dp.make_params(rows, rows)
    batch = dp.DpBatch(inputs, params, tb_budget_bytes=budget_gib << 30)
    chunks, order = batch.chunks()
    if budget_gib:
        assert len(chunks) - 1 >= 3, "this budget must split the batch into several chunks"
    batch.run(traceback=True)
    scores, ops, n_ops = batch.fetch()
    paths = batch.paths(ops, n_ops)
    sample = []
    for c in range(len(chunks) - 1):
        sample += [order[chunks[c]], order[chunks[c + 1] - 1]]
    sample += list(np.random.default_rng(1).integers(0, n, size=6))
    check_sample(inputs, params, scores, paths, sample, full_paths_for=[order[chunks[1] - 1], order[min(chunks[1], n - 1)]])
    assert all(len(p) >= L for p in paths[::97])
    # score-only pass: same scores
    batch.run(traceback=False)
    s2, _, _ = batch.fetch(with_paths=False)
    assert np.array_equal(s2, scores)
    batch.close()


@pytest.mark.parametrize("n_pairs", [64, 128, 512])
def test_deep_profiles_32_rows_10k_columns(n_pairs, oracle_build):
    """BASELINE.json configs[4]'s shape (32 rows x 10 kbp, int16 column weights) at 64 / 128 / 512 pairs: the launches that take
    the 8-, 4- and 2-wave stripe pipelines.  fetch() fails loudly if a stripe ever timed out waiting for its neighbour."""
    rows, L = 32, 10000
    inputs = dp.synth_batch(77 + n_pairs, np.full(n_pairs, L), np.full(n_pairs, L), rows, rows)
    params = dp.make_params(rows, rows)
    batch = dp.DpBatch(inputs, params)
    assert not batch.variant()["dot4"]
    batch.run(traceback=True)
    scores, ops, n_ops = batch.fetch()  # raises on pipe_error != 0
    paths = batch.paths(ops, n_ops)
    check_sample(inputs, params, scores, paths, [0, 1, n_pairs // 2, n_pairs - 1])
    batch.close()


@pytest.mark.parametrize("n_pairs,rows,L", [(8, 32, 10000), (48, 32, 10000), (100, 8, 4096)])
def test_pairs_split_over_workgroups_repeat_exactly(n_pairs, rows, L, oracle_build, monkeypatch):
    """Fewer pairs than CUs: every pair's stripes run on several workgroups that hand the seam over through agent-scope atomics.
    Twenty passes over the same batch must give the same scores and the same ops, byte for byte, as the one-workgroup-per-pair
    launch (PM_DP_GROUPS=1), in both path modes; a sample is checked against the oracle."""
    inputs = dp.synth_batch(900 + n_pairs, np.full(n_pairs, L), np.full(n_pairs, L), rows, rows)
    params = dp.make_params(rows, rows)
    for mode in ("ckpt", "bits"):
        monkeypatch.setenv("PM_DP_MODE", mode)
        monkeypatch.setenv("PM_DP_GROUPS", "1")
        ref = dp.DpBatch(inputs, params)
        ref.run(traceback=True)
        r_scores, r_ops, r_n = ref.fetch()
        ref.close()
        monkeypatch.delenv("PM_DP_GROUPS")
        batch = dp.DpBatch(inputs, params)
        for _ in range(20):
            batch.run(traceback=True)
            scores, ops, n_ops = batch.fetch()
            assert np.array_equal(scores, r_scores) and np.array_equal(n_ops, r_n)
            p0, p1 = batch.paths(ops, n_ops), dp.paths_of(inputs, r_ops, r_n)
            assert all(np.array_equal(a, b) for a, b in zip(p0, p1))
        if mode == "ckpt":
            check_sample(inputs, params, scores, batch.paths(ops, n_ops), [0, n_pairs - 1])
        batch.close()


@pytest.mark.parametrize("n,budget_gib", [(20000, 8), (100000, 96)])
def test_ragged_segment_batch_stand_in_for_config_2(n, budget_gib, oracle_build):
    """Stand-in for BASELINE.json configs[2] ("~100 k segment profile alignments"; nucmer is not in the image): ragged 4-row
    pairs, lengths log-normal (median 1 500, sigma 0.6, clipped to [200, 8 000]), lb = la * (1 + N(0, 0.05)) -- the full 100 000
    pairs with a 96 GiB workspace (chunks of a third of that, their fill kernels overlapping on three streams) and 20 000 pairs
    with 8 GiB; the batch at its default budget and one GPU's eighth of it (the tiers) are checked exhaustively in
    tests/test_dp_full_gpu.py.  The batch is cut into at least three workspace chunks and the sample holds the pairs either side of
    every chunk border."""
    rows = 4
    la, lb = dp.ragged_lengths(20261003, n)
    if n > 20000:
        from paramugsy_amd.synth_device import synth_batch_device
        inputs = synth_batch_device(20261003, la, lb, rows, rows, device="cuda")
    else:
        inputs = dp.synth_batch(20261003, la, lb, rows, rows)
    params = dp.make_params(rows, rows)
    batch = dp.DpBatch(inputs, params, tb_budget_bytes=budget_gib << 30)
    chunks, porder = batch.chunks()
    assert len(chunks) - 1 >= (3 if budget_gib else 1)
    assert sorted(porder.tolist()) == list(range(n))  # a permutation: longest pairs first
    batch.run(traceback=True)
    scores, ops, n_ops = batch.fetch()
    paths = batch.paths(ops, n_ops)
    order = np.argsort(la * lb)
    sample = [0, n - 1, order[0], order[-1], order[n // 2], porder[0], porder[-1], porder[100], porder[300], porder[600], porder[1500], porder[2500]]
    for c in range(1, len(chunks) - 1):
        sample += [porder[chunks[c] - 1], porder[chunks[c]]]
    small = [int(k) for k in order[:3]] + [int(order[n // 2])]
    check_sample(inputs, params, scores, paths, sample, full_paths_for=small)
    assert any((p != 0).any() for p in paths[:50])  # unequal lengths: the optimal paths carry gaps
    # every path spans its pair: as many ops as consume la columns of A and lb of B
    for k in range(0, n, max(1, n // 500)):
        assert (paths[k] != 1).sum() == la[k] and (paths[k] != 2).sum() == lb[k]
    batch.close()


def test_scores_beyond_the_int32_headroom_are_refused():
    """100-row x 10 kbp profiles with match 5 could push the skewed scores past 2^28: refused at creation (PM_E_INVALID)."""
    from paramugsy_amd import capi
    n, L = 2, 10000
    cols = np.zeros((n * L, 8), dtype=np.uint8)
    cols[:, 0] = 100
    off = np.arange(n + 1, dtype=np.int64) * L
    with pytest.raises(capi.PmError) as e:
        dp.DpBatch(dp.DpInputs(cols, off, cols.copy(), off.copy()), dp.make_params(1, 1, open_per_pair=40, extend_per_pair=3))
    assert e.value.code == capi.PM_E_INVALID
