"""BASELINE.json's DP configurations WHOLE, checked exhaustively: for the headline batch (100 000 pairs of 8 rows
x 4 096 columns), the ragged stand-in for configs[2] at 100 000 pairs, configs[1] at its stated 10 000 pairs and configs[4] at its
4 096 deep pairs

  * EVERY score against oracle/dp_tuned.c on all host cores (the tuned scorer is itself held to oracle/dp_oracle.c by
    tests/test_dp_oracle.py);
  * EVERY path re-scored under the specification (oracle/dp_oracle.c, dp_oracle_score_of_paths) and checked to span its pair -- a path
    that spans the pair and re-scores to the optimal score is an optimal alignment;
  * the oracle's own path, op for op, on at least 100 pairs spread over every workspace chunk and every tier of the launches
    (the first, the last and evenly spaced positions of every chunk in processing order; the tiers are the longest pairs, at the
    head of a chunk).

The oracle is this repo's own specification (the reference has no DP): "parity unpinned"."""
import numpy as np
import pytest

from paramugsy_amd import dp

pytestmark = pytest.mark.gpu


def spread_over_chunks(chunks, order, at_least=100):
    """Pairs at the first, the last and evenly spaced positions of every chunk (and around positions 256 and 768 of a chunk, where the
    tiers of a ragged launch end), in processing order: at least `at_least` of them."""
    n_chunks = len(chunks) - 1
    per_chunk = max(8, -(-at_least // n_chunks))
    picked = []
    for c in range(n_chunks):
        lo, hi = int(chunks[c]), int(chunks[c + 1])
        if hi <= lo:
            continue
        pos = set(int(x) for x in np.linspace(lo, hi - 1, per_chunk))
        pos |= {lo, hi - 1}
        pos |= {p for p in (lo + 1, lo + 255, lo + 256, lo + 767, lo + 768) if p < hi}
        picked += [int(order[p]) for p in sorted(pos)]
    return sorted(set(picked))


def check_whole_batch(inputs, params, budget_gib=0, min_chunks=1, oracle_pairs=100):
    import pyoracle
    batch = dp.DpBatch(inputs, params, tb_budget_bytes=budget_gib << 30)
    chunks, order = batch.chunks()
    assert len(chunks) - 1 >= min_chunks
    assert sorted(order.tolist()) == list(range(inputs.n_pairs))
    batch.run(traceback=True)
    scores, ops, n_ops = batch.fetch()
    variant = batch.variant()
    batch.close()
    bad_scores, bad_paths = pyoracle.dp_check_batch_exhaustively(inputs, params, scores, ops, n_ops)
    assert len(bad_scores) == 0, "scores differ from the tuned CPU scorer on pairs %s ..." % bad_scores[:8]
    assert len(bad_paths) == 0, "paths that do not span their pair or do not re-score to the reported score: pairs %s ..." % bad_paths[:8]
    sample = spread_over_chunks(chunks, order, oracle_pairs)
    assert len(sample) >= min(oracle_pairs, inputs.n_pairs)
    paths = dp.paths_of(inputs, ops, n_ops)
    for k, (o_score, o_path) in zip(sample, pyoracle.dp_align_pairs(inputs, params, sample)):
        assert o_score == scores[k] and np.array_equal(o_path, paths[k]), "oracle path of pair %d" % k
    return variant, len(chunks) - 1


def test_headline_batch_100k_pairs_of_8_rows_by_4096_exhaustively(oracle_build):
    """The batch the north-star target is quoted on, whole, on one GPU (6.5 GB of packed columns drawn on the GPU; the path workspace in
    chunks whose fill kernels overlap): 100 000 scores, 100 000 paths, the oracle's path on pairs of every chunk."""
    from paramugsy_amd.synth_device import synth_batch_device
    n, rows, L = 100000, 8, 4096
    inputs = synth_batch_device(20261003 * 1000003, np.full(n, L), np.full(n, L), rows, rows, device="cuda")
    variant, n_chunks = check_whole_batch(inputs, dp.make_params(rows, rows), min_chunks=3)
    assert variant["checkpoints"] and variant["cols_per_lane"] == 16


def test_ragged_100k_stand_in_for_config_2_exhaustively(oracle_build):
    """The stand-in for configs[2] / [3] at its 100 000 ragged pairs (narrow last stripes, the automatic cut in two chunks)."""
    from paramugsy_amd.synth_device import synth_batch_device
    n, rows = 100000, 4
    la, lb = dp.ragged_lengths(20261003, n)
    inputs = synth_batch_device(20261003, la, lb, rows, rows, device="cuda")
    check_whole_batch(inputs, dp.make_params(rows, rows))


def test_ragged_eighth_with_tiers_exhaustively(oracle_build):
    """One GPU's eighth of the ragged batch (configs[3]'s per-GPU share): the launch its longest pairs bound -- tiers of several
    wavefronts per pair with path kernels of their own beside the launch of the rest -- every score and every path."""
    n, rows = 12500, 4
    la, lb = dp.ragged_lengths(20261003, n)
    inputs = dp.synth_batch(20261003, la, lb, rows, rows)
    check_whole_batch(inputs, dp.make_params(rows, rows))


def test_config_1_at_its_10000_pairs_exhaustively(oracle_build):
    """BASELINE.json configs[1] as stated: 10 000 pairs of 2 rows x 1 000 columns."""
    inputs = dp.synth_pairs_fast(20261003, 10000, 2, 1000)
    check_whole_batch(inputs, dp.make_params(2, 2))


def test_deep_profiles_4096_pairs_of_32_rows_by_10k_exhaustively(oracle_build):
    """BASELINE.json configs[4] whole: 4 096 pairs of 32 rows x 10 000 columns (int16 column weights; 108 GB of checkpoints)."""
    from paramugsy_amd.synth_device import synth_batch_device
    n, rows, L = 4096, 32, 10000
    inputs = synth_batch_device(20261003 * 1000003 + 7, np.full(n, L), np.full(n, L), rows, rows, device="cuda")
    variant, _ = check_whole_batch(inputs, dp.make_params(rows, rows), oracle_pairs=100)
    assert not variant["dot4"]
