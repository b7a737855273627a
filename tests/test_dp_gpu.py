"""Profile x profile DP on the GPU against this repo's scalar oracle (oracle/dp_oracle.c).
NO REFERENCE COUNTERPART exists for this computation: "parity" here means kernel == this repo's own
specification, bit-exact on int32 scores and on every traceback op."""
import numpy as np
import pytest

from paramugsy_amd import dp
from paramugsy_amd.shard import slice_pairs

pytestmark = pytest.mark.gpu


def run_and_compare(inputs, params, tb_budget=0):
    """Scores and paths of the batch against the oracle's full-matrix aligner -- with the paths taken from checkpoints AND from
    stored decision bits (a batch as small as the tests' would choose the bits by itself, so both are forced in turn), unless the
    calling test has fixed PM_DP_MODE itself."""
    import os
    import pyoracle
    o_scores, o_paths = pyoracle.dp_align(inputs, params)
    preset = os.environ.get("PM_DP_MODE")
    out = None
    for mode in ([preset] if preset else ["ckpt", "bits"]):
        os.environ["PM_DP_MODE"] = mode
        try:
            batch = dp.DpBatch(inputs, params, tb_budget_bytes=tb_budget)
        finally:
            if preset is None:
                del os.environ["PM_DP_MODE"]
        assert batch.variant()["checkpoints"] == (mode == "ckpt")
        batch.run(traceback=True)
        scores, ops, n_ops = batch.fetch()
        assert np.array_equal(scores, o_scores), mode
        paths = batch.paths(ops, n_ops)
        for k in range(inputs.n_pairs):
            assert np.array_equal(paths[k], o_paths[k]), "pair %d (%s)" % (k, mode)
        # score-only pass gives the same scores
        batch.run(traceback=False)
        s2, _, _ = batch.fetch(with_paths=False)
        assert np.array_equal(s2, o_scores)
        info = batch.info()
        batch.close()
        out = (scores, paths, info)
    return out


@pytest.mark.parametrize("rows,length", [(2, 100), (3, 700), (8, 513), (32, 300)])
def test_equal_length_pairs(rows, length, oracle_build):
    inputs = dp.synth_pairs(1000 + rows, 12, rows, length)
    run_and_compare(inputs, dp.make_params(rows, rows))


def test_ragged_lengths_and_stripe_edges(oracle_build):
    # lengths around the stripe width (512 columns of B) and the LDS ring (64/128 rows of A), plus tiny profiles
    rng = np.random.default_rng(5)
    la = [1, 1, 2, 63, 64, 65, 127, 128, 129, 200, 511, 512, 513, 1025, 37, 300]
    lb = [1, 5, 1, 512, 511, 513, 64, 1, 1024, 1030, 77, 512, 513, 40, 1500, 300]
    cols = lambda n: np.concatenate([rng.integers(0, 4, size=(n, 5)).astype(np.uint8), np.zeros((n, 3), np.uint8)], axis=1)
    A = [cols(n) for n in la]
    B = [cols(n) for n in lb]
    inputs = dp.DpInputs(np.concatenate(A), np.concatenate([[0], np.cumsum(la)]).astype(np.int64), np.concatenate(B),
                         np.concatenate([[0], np.cumsum(lb)]).astype(np.int64))
    run_and_compare(inputs, dp.make_params(3, 3))


def test_empty_profiles(oracle_build):
    rng = np.random.default_rng(6)
    la = [0, 5, 0, 7]
    lb = [4, 0, 0, 7]
    cols = lambda n: np.concatenate([rng.integers(0, 3, size=(n, 5)).astype(np.uint8), np.zeros((n, 3), np.uint8)], axis=1)
    inputs = dp.DpInputs(np.concatenate([cols(n) for n in la]), np.concatenate([[0], np.cumsum(la)]).astype(np.int64),
                         np.concatenate([cols(n) for n in lb]), np.concatenate([[0], np.cumsum(lb)]).astype(np.int64))
    scores, paths, _ = run_and_compare(inputs, dp.make_params(2, 2))
    assert [len(p) for p in paths] == [4, 5, 0, len(paths[3])]


def test_varying_lengths_with_indels(oracle_build):
    inputs = dp.synth_pairs(77, 40, 4, 400, indel_rate=0.03, vary_length=True)
    scores, paths, _ = run_and_compare(inputs, dp.make_params(4, 4))
    assert any((p != 0).any() for p in paths)  # some optimal paths carry gaps


@pytest.mark.parametrize("mode", ["bits", "ckpt"])
def test_chunked_workspace_gives_same_results(mode, oracle_build, monkeypatch):
    monkeypatch.setenv("PM_DP_MODE", mode)
    inputs = dp.synth_pairs(78, 30, 2, 600)
    params = dp.make_params(2, 2)
    s1, p1, i1 = run_and_compare(inputs, params)
    s2, p2, i2 = run_and_compare(inputs, params, tb_budget=3 << 20)  # forces several chunks
    assert i1["chunks"] == 1 and i2["chunks"] > 3
    assert np.array_equal(s1, s2) and all(np.array_equal(a, b) for a, b in zip(p1, p2))


@pytest.mark.parametrize("slots", ["2", "3", "5"])
@pytest.mark.parametrize("one_stream", [False, True])
def test_chunks_on_overlapping_fill_streams_repeat_exactly(slots, one_stream, oracle_build, monkeypatch):
    """A batch of several chunks: the workspace in 2, 3 or 5 parts, the fill kernels of every part on a stream of their own
    (chunk c + 1's kernel starts while chunk c's drains; it waits only for the path kernel that frees its part) or all on the
    caller's stream -- the oracle's scores and paths whichever way, run after run on the same batch, and on a caller's stream
    that is not the null stream."""
    import pyoracle
    import torch
    monkeypatch.setenv("PM_DP_SLOTS", slots)
    if one_stream:
        monkeypatch.setenv("PM_DP_ONE_FILL_STREAM", "1")
    inputs = dp.synth_pairs(781, 120, 4, 700, indel_rate=0.02, vary_length=True)
    params = dp.make_params(4, 4)
    o_scores, o_paths = pyoracle.dp_align(inputs, params)
    batch = dp.DpBatch(inputs, params, tb_budget_bytes=6 << 20)
    assert batch.info()["chunks"] >= 2 * int(slots)
    side = torch.cuda.Stream()
    for rep in range(4):
        stream = side.cuda_stream if rep & 1 else 0
        if rep == 3:
            ms_fill, ms_path = batch.run_profiled(True, stream)
            assert 0 < batch.fill_busy_ms() <= ms_fill * 1.001
        else:
            batch.run(True, stream)
        scores, ops, n_ops = batch.fetch()
        assert np.array_equal(scores, o_scores), rep
        paths = batch.paths(ops, n_ops)
        assert all(np.array_equal(a, b) for a, b in zip(paths, o_paths)), rep
    batch.close()


@pytest.mark.parametrize("gate", [True, False])
def test_batch_that_fits_cut_into_chunks_with_parts_of_their_own(gate, oracle_build, monkeypatch):
    """PM_DP_SPLIT=N cuts a batch whose workspace fits the budget into N chunks, every one with its own part of the workspace (no
    part is reused, so no fill kernel waits for a path kernel); with and without the gate kernel in front of the fill kernels."""
    import pyoracle
    monkeypatch.setenv("PM_DP_SPLIT", "5")
    if not gate:
        monkeypatch.setenv("PM_DP_NO_GATE", "1")
    inputs = dp.synth_pairs(782, 90, 4, 650, indel_rate=0.02, vary_length=True)
    params = dp.make_params(4, 4)
    o_scores, o_paths = pyoracle.dp_align(inputs, params)
    batch = dp.DpBatch(inputs, params)
    assert batch.info()["chunks"] in (4, 5, 6)
    for rep in range(3):
        batch.run(True)
        scores, ops, n_ops = batch.fetch()
        assert np.array_equal(scores, o_scores), rep
        assert all(np.array_equal(a, b) for a, b in zip(batch.paths(ops, n_ops), o_paths)), rep
    batch.close()


@pytest.mark.parametrize("mode", ["bits", "ckpt"])
def test_few_long_pairs_in_chunks_on_overlapping_streams_keep_one_workgroup_per_pair(mode, oracle_build, monkeypatch):
    """Found by tools/dp_fuzz.py (seed 900234): two long pairs cut into two chunks whose fill launches overlap, each launch asked for
    several workgroups per pair -- which need every workgroup of the launch resident and share the batch's one set of progress
    words.  Launches that may overlap keep one workgroup per pair."""
    import pyoracle
    for k, v in (("PM_DP_MODE", mode), ("PM_DP_WAVES", "4"), ("PM_DP_GROUPS", "8"), ("PM_DP_SLOTS", "4"), ("PM_DP_SPLIT", "5")):
        monkeypatch.setenv(k, v)
    inputs = dp.synth_batch(900234, np.array([4213, 4534]), np.array([1424, 1399]), 8, 8)
    params = dp.make_params(8, 8)
    o_scores, o_paths = pyoracle.dp_align(inputs, params)
    batch = dp.DpBatch(inputs, params)
    assert batch.info()["chunks"] == 2
    for rep in range(3):
        batch.run(True)
        scores, ops, n_ops = batch.fetch()
        assert np.array_equal(scores, o_scores), rep
        assert all(np.array_equal(a, b) for a, b in zip(batch.paths(ops, n_ops), o_paths)), rep
    batch.close()


@pytest.mark.parametrize("waves", ["1", "2", "4", "8", "16"])
@pytest.mark.parametrize("cols", ["8", "16"])
@pytest.mark.parametrize("dot4", ["0", "1"])
@pytest.mark.parametrize("mode", ["bits", "ckpt"])
def test_every_kernel_variant_on_multi_stripe_pairs(waves, cols, dot4, mode, oracle_build, monkeypatch):
    """One wave per pair and the 2-, 4- and 8-wave stripe pipelines, 8 and 16 columns per lane, int8 and int16 column scores,
    paths from stored decision bits and from checkpoints: all must give the oracle's scores and paths on pairs that span
    several stripes (B up to 2 600 columns)."""
    monkeypatch.setenv("PM_DP_WAVES", waves)
    monkeypatch.setenv("PM_DP_COLS", cols)
    monkeypatch.setenv("PM_DP_DOT4", dot4)
    monkeypatch.setenv("PM_DP_MODE", mode)
    rng = np.random.default_rng(int(waves) * 100 + int(cols) + int(dot4))
    la = [700, 64, 1300, 129, 2000, 5, 150]
    lb = [2600, 1025, 1024, 2049, 513, 1100, 9300 if cols == "8" else 12500]  # the last: 19 / 13 stripes, more than 16 waves
    cols_of = lambda n: np.concatenate([rng.integers(0, 3, size=(n, 5)).astype(np.uint8), np.zeros((n, 3), np.uint8)], axis=1)
    inputs = dp.DpInputs(np.concatenate([cols_of(n) for n in la]), np.concatenate([[0], np.cumsum(la)]).astype(np.int64),
                         np.concatenate([cols_of(n) for n in lb]), np.concatenate([[0], np.cumsum(lb)]).astype(np.int64))
    run_and_compare(inputs, dp.make_params(4, 4))


@pytest.mark.parametrize("groups,waves", [("2", "2"), ("4", "2"), ("2", "16"), ("8", "4"), ("16", "2"), ("4", "0")])
@pytest.mark.parametrize("cols", ["8", "16"])
@pytest.mark.parametrize("mode", ["bits", "ckpt"])
def test_stripes_of_a_pair_on_several_workgroups(groups, waves, cols, mode, oracle_build, monkeypatch):
    """A pair's stripes dealt to the waves of several workgroups (few pairs: fewer workgroups than CUs otherwise), the seam
    handed over through global memory with agent-scope release / acquire: same scores and paths as the oracle, for teams
    smaller and larger than the number of stripes, and for the split dp_launch_fill picks by itself (waves "0")."""
    monkeypatch.setenv("PM_DP_GROUPS", groups)
    if waves != "0":
        monkeypatch.setenv("PM_DP_WAVES", waves)
    monkeypatch.setenv("PM_DP_COLS", cols)
    monkeypatch.setenv("PM_DP_MODE", mode)
    rng = np.random.default_rng(int(groups) * 100 + int(waves) + int(cols))
    la = [700, 64, 1300, 129, 900, 5, 150]
    lb = [2600, 1025, 4100, 2049, 513, 1100, 9300 if cols == "8" else 17000]
    cols_of = lambda n: np.concatenate([rng.integers(0, 3, size=(n, 5)).astype(np.uint8), np.zeros((n, 3), np.uint8)], axis=1)
    inputs = dp.DpInputs(np.concatenate([cols_of(n) for n in la]), np.concatenate([[0], np.cumsum(la)]).astype(np.int64),
                         np.concatenate([cols_of(n) for n in lb]), np.concatenate([[0], np.cumsum(lb)]).astype(np.int64))
    run_and_compare(inputs, dp.make_params(4, 4))


@pytest.mark.parametrize("lanes,cols", [("8", "16"), ("16", "16"), ("32", "16"), ("64", "16"), ("4", "8"), ("8", "8"), ("16", "8"), ("32", "8")])
def test_checkpoint_walk_with_every_group_size(lanes, cols, oracle_build, monkeypatch):
    """The checkpoint walk with every group size (lanes per pair) the block width allows, for both column counts of the fill
    kernel: ragged pairs whose optimal paths carry long gaps, so the walk leaves blocks through their left edge as well as
    through their top."""
    monkeypatch.setenv("PM_DP_MODE", "ckpt")
    monkeypatch.setenv("PM_DP_WALK_LANES", lanes)
    monkeypatch.setenv("PM_DP_COLS", cols)
    inputs = dp.synth_pairs(300 + int(lanes), 37, 3, 500, indel_rate=0.03, vary_length=True)
    run_and_compare(inputs, dp.make_params(3, 3))
    rng = np.random.default_rng(int(lanes))
    la = [900, 40, 1, 33, 1200]
    lb = [60, 1300, 700, 32, 1100]
    cols_of = lambda n: np.concatenate([rng.integers(0, 3, size=(n, 5)).astype(np.uint8), np.zeros((n, 3), np.uint8)], axis=1)
    inputs = dp.DpInputs(np.concatenate([cols_of(n) for n in la]), np.concatenate([[0], np.cumsum(la)]).astype(np.int64),
                         np.concatenate([cols_of(n) for n in lb]), np.concatenate([[0], np.cumsum(lb)]).astype(np.int64))
    run_and_compare(inputs, dp.make_params(3, 3))


def shifted_pairs(seed, rows):
    """Pairs whose optimal paths run far from the straight line between the corners: the two profiles share a long core that
    sits at different offsets in them (junk before it in one, after it in the other, or in the middle)."""
    rng = np.random.default_rng(seed)

    def one_hot(seq):
        c = np.zeros((len(seq), 8), np.uint8)
        c[np.arange(len(seq)), seq] = rows
        return c
    A, B = [], []
    for core, junk, where in [(1200, 400, "ends"), (900, 300, "middle"), (700, 250, "ends"), (1500, 200, "middle"), (300, 500, "ends")]:
        s = rng.integers(0, 4, size=core)
        ja, jb = rng.integers(0, 4, size=junk), rng.integers(0, 4, size=junk)
        if where == "ends":  # B = junk + core, A = core + junk: the path is `junk` columns off the diagonal all along
            A.append(one_hot(np.concatenate([s, ja])))
            B.append(one_hot(np.concatenate([jb, s])))
        else:  # the first half on the diagonal, then B carries an insertion that A makes up for at its end
            h = core // 2
            A.append(one_hot(np.concatenate([s, ja])))
            B.append(one_hot(np.concatenate([s[:h], jb, s[h:]])))
    la, lb = [len(x) for x in A], [len(x) for x in B]
    return dp.DpInputs(np.concatenate(A), np.concatenate([[0], np.cumsum(la)]).astype(np.int64), np.concatenate(B),
                       np.concatenate([[0], np.cumsum(lb)]).astype(np.int64))


@pytest.mark.parametrize("band", ["0", "1"])
@pytest.mark.parametrize("lanes,cols", [("0", "16"), ("8", "16"), ("64", "16"), ("4", "8"), ("32", "8")])
def test_walk_band_and_paths_that_leave_it(lanes, cols, band, oracle_build, monkeypatch):
    """The checkpoint walk reads the blocks around the straight corner-to-corner line from the band computed up front and
    recomputes the others: paths that stay in the band, paths hundreds of columns away from it, and paths that enter and leave
    it must all equal the oracle's, with the band on (forced) and off."""
    monkeypatch.setenv("PM_DP_MODE", "ckpt")
    monkeypatch.setenv("PM_DP_BAND", band)
    monkeypatch.setenv("PM_DP_COLS", cols)
    if lanes != "0":
        monkeypatch.setenv("PM_DP_WALK_LANES", lanes)
    inputs = shifted_pairs(11 + int(lanes), 3)
    scores, paths, _ = run_and_compare(inputs, dp.make_params(3, 3))
    # the paths do run far from the line: at least 150 rows off somewhere
    far = 0
    for k, path in enumerate(paths):
        la = int(inputs.off_a[k + 1] - inputs.off_a[k])
        lb = int(inputs.off_b[k + 1] - inputs.off_b[k])
        i = np.cumsum(path != 1)
        j = np.cumsum(path != 2)
        far = max(far, int(np.abs(i - j * la / lb).max()))
    assert far >= 150
    # and pairs that stay near it
    inputs = dp.synth_pairs(500 + int(lanes), 20, 3, 900, indel_rate=0.02, vary_length=True)
    run_and_compare(inputs, dp.make_params(3, 3))


@pytest.mark.parametrize("rows,expect_dot4", [(25, True), (26, False), (200, False)])
def test_int8_path_selection_at_its_boundary(rows, expect_dot4, oracle_build):
    """rows x max|sub| = 125 fits int8 (dot4 path), 130 does not (int16 path); 200-row columns exercise counts above
    127 on the int16 path.  Both must equal the oracle."""
    rng = np.random.default_rng(rows)
    n, L = 6, 150
    def cols():
        c = np.zeros((n * L, 8), dtype=np.uint8)
        pick = rng.integers(0, 5, size=(n * L, rows))
        for s in range(5):
            c[:, s] = (pick == s).sum(axis=1)
        c[::7, :5] = 0
        c[::7, rng.integers(0, 4)] = rows  # whole columns of one base: the extreme weights
        return c
    off = np.arange(n + 1, dtype=np.int64) * L
    inputs = dp.DpInputs(cols(), off, cols(), off.copy())
    params = dp.make_params(1, 1, open_per_pair=40 * rows, extend_per_pair=3 * rows)
    batch = dp.DpBatch(inputs, params)
    assert batch.variant()["dot4"] == expect_dot4
    batch.close()
    run_and_compare(inputs, params)


@pytest.mark.parametrize("dot4", ["1", "0"])
@pytest.mark.parametrize("mode", ["ckpt", "bits"])
def test_uniform_depth_variant_and_its_fallbacks(dot4, mode, oracle_build, monkeypatch):
    """Every column of A holding the same number of symbols (rows of a MAF block without N's) selects the kernels that fold the
    gap row of the score into the base weights; one column with an N (counted in byte 5, so bytes 0-4 sum to one less) or
    PM_DP_UNI=0 selects the general kernels.  All equal the oracle."""
    monkeypatch.setenv("PM_DP_DOT4", dot4)
    monkeypatch.setenv("PM_DP_MODE", mode)
    inputs = dp.synth_pairs(61, 14, 4, 1300, indel_rate=0.02, vary_length=True)  # B up to two stripes
    params = dp.make_params(4, 4)
    b = dp.DpBatch(inputs, params)
    assert b.variant()["uniform_depth"] and b.variant()["dot4"] == (dot4 == "1")
    b.close()
    run_and_compare(inputs, params)
    monkeypatch.setenv("PM_DP_UNI", "0")
    b = dp.DpBatch(inputs, params)
    assert not b.variant()["uniform_depth"]
    b.close()
    run_and_compare(inputs, params)
    monkeypatch.delenv("PM_DP_UNI")
    cols_a = inputs.cols_a.copy()
    cols_a[7, 0:4] = [1, 1, 1, 0]
    cols_a[7, 4] = 0
    cols_a[7, 5] = 1  # an N: neither base nor gap
    ragged = dp.DpInputs(cols_a, inputs.off_a, inputs.cols_b, inputs.off_b)
    b = dp.DpBatch(ragged, params)
    assert not b.variant()["uniform_depth"]
    b.close()
    run_and_compare(ragged, params)
    # A uniform, B not: still the uniform kernels (only A's depth matters)
    cols_b = inputs.cols_b.copy()
    cols_b[3, 4] = 0
    half = dp.DpInputs(inputs.cols_a, inputs.off_a, cols_b, inputs.off_b)
    b = dp.DpBatch(half, params)
    assert b.variant()["uniform_depth"]
    b.close()
    run_and_compare(half, params)


def test_path_mode_is_chosen_by_batch_size(oracle_build):
    """A batch of a few short pairs stores decision bits (the checkpoint walk's chain of blocks and its extra launches have a
    latency that does not shrink with the batch), as does a mid-size one when the walk has no band to read (PM_DP_BAND=0); a
    large one leaves checkpoints; PM_DP_MODE fixes it either way (the other tests do)."""
    params = dp.make_params(2, 2)
    small = dp.DpBatch(dp.synth_pairs_fast(1, 64, 2, 150), params)
    assert not small.variant()["checkpoints"]
    small.close()
    import os
    params8 = dp.make_params(8, 8)
    mid_in = dp.synth_pairs_fast(1, 64, 8, 4096)
    mid = dp.DpBatch(mid_in, params8)
    assert mid.variant()["checkpoints"]
    mid.close()
    os.environ["PM_DP_BAND"] = "0"
    try:
        mid = dp.DpBatch(mid_in, params8)
    finally:
        del os.environ["PM_DP_BAND"]
    assert not mid.variant()["checkpoints"]
    mid.close()
    big_in = dp.synth_pairs_fast(2, 6000, 2, 1000)
    big = dp.DpBatch(big_in, params)
    assert big.variant()["checkpoints"]
    big.run(True)
    scores, ops, n_ops = big.fetch()
    big.close()
    import pyoracle
    assert np.array_equal(scores, pyoracle.dp_scores(big_in, params))
    for k in range(0, 6000, 997):
        rc, s_ = pyoracle.dp_score_of_path(big_in, params, k, dp.paths_of(big_in, ops, n_ops)[k])
        assert rc == 0 and s_ == scores[k]


def test_published_needleman_wunsch_example_on_the_gpu(oracle_build):
    from test_dp_oracle import needleman_wunsch_textbook_case
    inputs, params = needleman_wunsch_textbook_case()
    scores, paths, _ = run_and_compare(inputs, params)
    assert scores.tolist() == [0]


def test_weights_beyond_int16_are_refused():
    from paramugsy_amd import capi
    cols = np.zeros((4, 8), dtype=np.uint8)
    cols[:, :5] = 255
    off = np.array([0, 4], dtype=np.int64)
    p = dp.make_params(1, 1, match=127, mismatch=-127)
    with pytest.raises(capi.PmError) as e:
        dp.DpBatch(dp.DpInputs(cols, off, cols.copy(), off.copy()), p)
    assert e.value.code == capi.PM_E_INVALID


def test_general_substitution_matrix_and_zero_penalties(oracle_build):
    rng = np.random.default_rng(8)
    inputs = dp.synth_pairs(79, 10, 5, 250, vary_length=True)
    p = dp.make_params(5, 5)
    for k in range(25):
        p.sub[k] = int(rng.integers(-9, 10))
    run_and_compare(inputs, p)
    p.gap_open = 0
    p.gap_extend = 0
    run_and_compare(inputs, p)  # all-ties regime: exercises every tie-break rule


def test_full_size_properties_baseline_config_1(oracle_build):
    """BASELINE.json configs[1] at its stated 10 000 pairs of 2 rows x 1 kbp: every score equals the tuned CPU scorer's, every
    reported path re-scores to the reported score and spans its pair (the same batch with the oracle's paths op for op:
    tests/test_dp_full_gpu.py)."""
    import pyoracle
    inputs = dp.synth_pairs_fast(20261003, 10000, 2, 1000)
    params = dp.make_params(2, 2)
    batch = dp.DpBatch(inputs, params)
    batch.run(traceback=True)
    scores, ops, n_ops = batch.fetch()
    batch.close()
    bad_scores, bad_paths = pyoracle.dp_check_batch_exhaustively(inputs, params, scores, ops, n_ops)
    assert len(bad_scores) == 0 and len(bad_paths) == 0
    assert np.array_equal(scores[:300], pyoracle.dp_scores(slice_pairs(inputs, 0, 300), params))  # and the scalar oracle on the first 300


def test_big_profiles_8_rows_4k_columns(oracle_build):
    """The north-star shape (8 rows x 4 kbp), a few pairs: scores vs the oracle's scorer, paths by re-scoring."""
    import pyoracle
    inputs = dp.synth_pairs_fast(11, 6, 8, 4096)
    params = dp.make_params(8, 8)
    batch = dp.DpBatch(inputs, params)
    batch.run(traceback=True)
    scores, ops, n_ops = batch.fetch()
    assert np.array_equal(scores, pyoracle.dp_scores(inputs, params))
    for k, p in enumerate(batch.paths(ops, n_ops)):
        rc, s = pyoracle.dp_score_of_path(inputs, params, k, p)
        assert rc == 0 and s == scores[k]
    batch.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_shapes_random_scoring(seed, oracle_build, monkeypatch):
    """Fuzz: 150 pairs of unrelated random lengths (1..2 300 x 1..2 300: zero to three stripe seams, every tile and
    ring remainder), random 5x5 matrix and gap costs, random waves-per-pair choice.  Scores and paths equal the
    oracle's full-matrix aligner."""
    rng = np.random.default_rng(seed)
    monkeypatch.setenv("PM_DP_WAVES", ["1", "4", "8"][seed % 3])
    monkeypatch.setenv("PM_DP_MODE", ["ckpt", "bits", "ckpt"][seed % 3])
    n = 150
    la = rng.integers(1, 2300, size=n)
    lb = rng.integers(1, 2300, size=n)
    la[:6] = [1, 4, 63, 64, 65, 2299]
    lb[:6] = [2299, 1025, 1024, 1023, 1, 4]
    rows = int(rng.integers(1, 9))

    def cols(total):
        c = np.zeros((total, 8), dtype=np.uint8)
        pick = rng.integers(0, 5, size=(total, rows))
        for s in range(5):
            c[:, s] = (pick == s).sum(axis=1)
        return c

    inputs = dp.DpInputs(cols(int(la.sum())), np.concatenate([[0], np.cumsum(la)]).astype(np.int64), cols(int(lb.sum())),
                         np.concatenate([[0], np.cumsum(lb)]).astype(np.int64))
    p = dp.make_params(rows, rows)
    for k in range(25):
        p.sub[k] = int(rng.integers(-6, 7))
    p.gap_open = int(rng.integers(0, 40)) * rows
    p.gap_extend = int(rng.integers(0, 6)) * rows
    run_and_compare(inputs, p)


def test_options_at_creation_and_as_process_defaults_and_the_geometry_of_a_pass(oracle_build):
    """pm_dp_options_t given to pm_dp_batch_create_opt, and the process's defaults (pm_dp_set_default_options) for batches created
    without options; pm_dp_batch_geometry: the cells the stripes cover, narrow last stripes on or off, the fill launches of a pass.
    Every variant gives the oracle's scores and paths."""
    import ctypes as C
    import pyoracle
    from paramugsy_amd import capi
    la = [700, 1300, 90, 2100, 1024, 300]
    lb = [650, 1290, 100, 2300, 1025, 257]
    inputs = dp.synth_batch(5, la, lb, 3, 3)
    params = dp.make_params(3, 3)
    o_scores, o_paths = pyoracle.dp_align(inputs, params)
    cells = inputs.cells

    def check(batch):
        batch.run(True)
        scores, ops, n_ops = batch.fetch()
        assert np.array_equal(scores, o_scores)
        assert all(np.array_equal(p, q) for p, q in zip(batch.paths(ops, n_ops), o_paths))
    narrow = dp.DpBatch(inputs, params, options=dp.options(path_mode=2, cols_per_lane=16))
    full = dp.DpBatch(inputs, params, options=dp.options(path_mode=2, cols_per_lane=16, full_stripes=1))
    g_n, g_f = narrow.geometry(), full.geometry()
    assert g_n["narrow_last_stripes"] and not g_f["narrow_last_stripes"] and g_n["fill_launches"] == g_f["fill_launches"] == 1
    padded = lambda w: sum(a * ((b + w - 1) // w) * w for a, b in zip(la, lb))  # noqa: E731
    assert g_f["padded_cells"] == padded(1024) and cells <= g_n["padded_cells"] <= padded(256) < g_f["padded_cells"]
    check(narrow)
    check(full)
    narrow.close()
    full.close()
    with pytest.raises(capi.PmError):
        dp.DpBatch(inputs, params, options=dp.options(cols_per_lane=12))
    # the defaults of batches created without options: stored bits, then everything chosen again
    lib = capi.lib()
    ca, cb = np.ascontiguousarray(inputs.cols_a), np.ascontiguousarray(inputs.cols_b)
    oa, ob = np.ascontiguousarray(inputs.off_a, dtype=np.int64), np.ascontiguousarray(inputs.off_b, dtype=np.int64)

    def plain_create():
        h = C.c_void_p()
        capi.check(lib.pm_dp_batch_create(ca.ctypes.data, oa.ctypes.data, cb.ctypes.data, ob.ctypes.data, len(la), C.byref(params), 0, 0, C.byref(h)))
        ck = C.c_int32()
        capi.check(lib.pm_dp_batch_path_mode(h, C.byref(ck), None, None))
        lib.pm_dp_batch_destroy(h)
        return bool(ck.value)
    try:
        dp.set_default_options(dp.options(path_mode=2))
        assert plain_create() is True
        dp.set_default_options(dp.options(path_mode=1))
        assert plain_create() is False
    finally:
        dp.set_default_options(None)


@pytest.mark.parametrize("tile", ["64", "128", "320", "1024"])
@pytest.mark.parametrize("dot4", ["0", "1"])
@pytest.mark.parametrize("tail", ["0", "1"])
def test_tiles_from_a_queue_equal_the_stripe_kernel(tile, dot4, tail, oracle_build, monkeypatch):
    """dp_fill_tiles_kernel (round 5): the stripes of every pair cut into tiles of `tile` steps, taken from a queue by a persistent
    grid, the lanes' state handed from tile to tile through memory, the stripes of a pair pipelined through progress words -- scores,
    checkpoints (hence every path) and the scores-only pass equal the oracle's.  Pairs of one to five stripes, rows of A from less
    than one tile to a dozen, lengths around the tile and block edges, an empty profile on either side."""
    monkeypatch.setenv("PM_DP_MODE", "ckpt")
    monkeypatch.setenv("PM_DP_TILE", tile)
    monkeypatch.setenv("PM_DP_DOT4", dot4)
    monkeypatch.setenv("PM_DP_TAIL", tail)
    monkeypatch.setenv("PM_DP_COLS", "16")
    la = np.array([1, 63, 64, 65, 127, 128, 129, 191, 192, 193, 500, 700, 1000, 0, 300, 257, 640, 705])
    lb = np.array([2100, 1024, 1025, 3000, 5000, 100, 2049, 4097, 1023, 1500, 2600, 4200, 1300, 50, 0, 3073, 2048, 1100])
    inputs = dp.synth_batch(77, la, lb, 3, 4)
    run_and_compare(inputs, dp.make_params(3, 4))
    # the same tiles under the uniform-depth form and its general twin
    for uni in ("1", "0"):
        monkeypatch.setenv("PM_DP_UNI", uni)
        inputs = dp.synth_pairs(78, 6, 4, 900, vary_length=True)
        run_and_compare(inputs, dp.make_params(4, 4))


def test_tiles_are_what_a_launch_of_few_long_pairs_takes_by_itself_and_tiers_are_not(oracle_build, monkeypatch):
    """The rule (dp_tiles_rule): stripe-long jobs between half and eight times the chip's 4 096 wavefront slots, the longest pair two
    tiles or more.  600 pairs of 2 100 x 4 500 columns (five stripes each: 3 000 jobs) qualify; their results are checked through the
    size-independent properties (every path re-scores to its score and spans its pair) and against the stripe kernel (tile_steps = 1)."""
    import pyoracle
    monkeypatch.delenv("PM_DP_TILE", raising=False)
    inputs = dp.synth_batch(81, np.full(600, 2100), np.full(600, 4500), 2, 2)
    params = dp.make_params(2, 2)
    got = []
    for tile_steps in (0, 1):
        batch = dp.DpBatch(inputs, params, options=dp.options(path_mode=2, tile_steps=tile_steps))
        batch.run(traceback=True)
        scores, ops, n_ops = batch.fetch()
        got.append((scores, ops, n_ops))
        if tile_steps == 0:
            bad_scores, bad_paths = pyoracle.dp_check_batch_exhaustively(inputs, params, scores, ops, n_ops)
            assert len(bad_scores) == 0 and len(bad_paths) == 0
            paths = batch.paths(ops, n_ops)
            for pair, (o_score, o_path) in zip([0, 299, 599], pyoracle.dp_align_pairs(inputs, params, [0, 299, 599])):
                assert scores[pair] == o_score and np.array_equal(paths[pair], o_path)
        batch.close()
    assert np.array_equal(got[0][0], got[1][0]) and np.array_equal(got[0][2], got[1][2]) and np.array_equal(got[0][1], got[1][1])


@pytest.mark.parametrize("lanes", ["0", "16", "32"])
@pytest.mark.parametrize("dot4", ["0", "1"])
def test_walk_beside_the_fill_kernel_of_its_own_launch(lanes, dot4, oracle_build, monkeypatch):
    """The early walk (round 5; DpEarly in dp_internal.hpp): the fill wavefront that has finished a pair puts it on the list of the XCD
    it runs on, walkers launched BESIDE the fill kernel take the entries of their own XCD's list while the fill goes on, and a second
    launch of the walk behind the fill kernel takes what is left.  Forced for batches of any size (PM_DP_EARLY_WALK=1), band off: every
    score and path equal the oracle's -- ragged lengths (pairs finish at very different times), empty profiles, more pairs than early
    groups and fewer, several passes over the same batch (the lists are zeroed before every pass)."""
    import pyoracle
    monkeypatch.setenv("PM_DP_MODE", "ckpt")
    monkeypatch.setenv("PM_DP_EARLY_WALK", "1")
    monkeypatch.setenv("PM_DP_BAND", "0")
    monkeypatch.setenv("PM_DP_WAVES", "1")
    monkeypatch.setenv("PM_DP_DOT4", dot4)
    if lanes != "0":
        monkeypatch.setenv("PM_DP_WALK_LANES", lanes)
    la, lb = dp.ragged_lengths(91, 300, median=300, lo=1, hi=2400)
    la[7] = 0
    lb[100] = 0
    la[200] = lb[200] = 0
    inputs = dp.synth_batch(92, la, lb, 3, 3)
    params = dp.make_params(3, 3)
    o_scores, o_paths = pyoracle.dp_align(inputs, params)
    batch = dp.DpBatch(inputs, params)
    for rep in range(3):
        batch.run(True)
        scores, ops, n_ops = batch.fetch()
        assert np.array_equal(scores, o_scores), rep
        paths = batch.paths(ops, n_ops)
        bad = [k for k in range(inputs.n_pairs) if not np.array_equal(paths[k], o_paths[k])]
        assert not bad, (rep, bad[:5])
    batch.close()
    # a handful of pairs (fewer than the early walkers' groups), and the chunked pipeline with the early walk on every chunk
    run_and_compare(dp.synth_pairs(93, 5, 2, 700, vary_length=True), dp.make_params(2, 2))
    run_and_compare(dp.synth_batch(94, la[:120], lb[:120], 3, 3), params, tb_budget=6 << 20)
