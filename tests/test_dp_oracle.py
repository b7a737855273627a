"""The DP oracle against an independent numpy restatement of the same specification (small cases), and its own
internal consistency (full-matrix aligner vs two-row scorer vs path re-scoring).  CPU only.
"Parity unpinned": there is no reference implementation of this DP."""
import numpy as np
import pytest

from paramugsy_amd import dp

NEG = -(1 << 29)


def numpy_dp(a, b, p):
    """Direct transcription of the recurrence in oracle/dp_oracle.h with Python ints."""
    sub = np.array(list(p.sub), dtype=np.int64).reshape(5, 5)
    la, lb = len(a), len(b)
    go, ge = p.gap_open, p.gap_extend
    H = [[0] * (lb + 1) for _ in range(la + 1)]
    E = [[NEG] * (lb + 1) for _ in range(la + 1)]
    F = [[NEG] * (lb + 1) for _ in range(la + 1)]
    for j in range(1, lb + 1):
        H[0][j] = -(go + (j - 1) * ge)
    for i in range(1, la + 1):
        H[i][0] = -(go + (i - 1) * ge)
    for i in range(1, la + 1):
        for j in range(1, lb + 1):
            s = int(a[i - 1, :5].astype(np.int64) @ sub @ b[j - 1, :5].astype(np.int64))
            E[i][j] = max(E[i][j - 1] - ge, H[i][j - 1] - go)
            F[i][j] = max(F[i - 1][j] - ge, H[i - 1][j] - go)
            H[i][j] = max(H[i - 1][j - 1] + s, E[i][j], F[i][j])
    return H[la][lb]


@pytest.mark.parametrize("seed", range(6))
def test_oracle_scores_match_numpy_transcription(seed, oracle_build):
    import pyoracle
    inputs = dp.synth_pairs(seed, 4, 3, 25, indel_rate=0.05, vary_length=True)
    p = dp.make_params(3, 3)
    scores, paths = pyoracle.dp_align(inputs, p)
    fast = pyoracle.dp_scores(inputs, p)
    for k in range(inputs.n_pairs):
        a = inputs.cols_a[inputs.off_a[k]:inputs.off_a[k + 1]]
        b = inputs.cols_b[inputs.off_b[k]:inputs.off_b[k + 1]]
        assert scores[k] == numpy_dp(a, b, p) == fast[k]
        rc, s = pyoracle.dp_score_of_path(inputs, p, k, paths[k])
        assert rc == 0 and s == scores[k]
        assert (paths[k] != 1).sum() == len(a) and (paths[k] != 2).sum() == len(b)


def needleman_wunsch_textbook_case():
    """The worked example of the Needleman-Wunsch article most readers know (Wikipedia, "Needleman-Wunsch algorithm"): GCATGCG
    against GATTACA with match +1, mismatch -1, indel -1 has optimal global score 0.  One-row profiles and gap_open =
    gap_extend = 1 make this repo's profile DP that very recurrence."""
    a, b = dp.pack_profile([b"GCATGCG"]), dp.pack_profile([b"GATTACA"])
    inputs = dp.DpInputs(a, np.array([0, len(a)], np.int64), b, np.array([0, len(b)], np.int64))
    params = dp.make_params(1, 1, match=1, mismatch=-1, base_gap=-1, open_per_pair=1, extend_per_pair=1)
    return inputs, params


def test_published_needleman_wunsch_example(oracle_build):
    """The one known answer from outside this repo that the DP's specification can be held against (everything else about the DP
    is "parity unpinned": SURVEY.md 0)."""
    import pyoracle
    inputs, params = needleman_wunsch_textbook_case()
    scores, paths = pyoracle.dp_align(inputs, params)
    assert scores.tolist() == [0]
    assert (paths[0] != 1).sum() == 7 and (paths[0] != 2).sum() == 7
    # the path re-scores to 0 by hand: +1 per match, -1 per mismatch, -1 per gap column
    x, y, i, j, total = "GCATGCG", "GATTACA", 0, 0, 0
    for op in paths[0]:
        if op == 0:
            total += 1 if x[i] == y[j] else -1
            i, j = i + 1, j + 1
        elif op == 1:
            total, j = total - 1, j + 1
        else:
            total, i = total - 1, i + 1
    assert total == 0 and numpy_dp(a=inputs.cols_a, b=inputs.cols_b, p=params) == 0


def test_pack_profile_counts():
    cols = dp.pack_profile([b"ACGT-N", b"AC-TTa", b"-CGTAa"])
    assert cols[:, :5].tolist() == [[2, 0, 0, 0, 1], [0, 3, 0, 0, 0], [0, 0, 2, 0, 1], [0, 0, 0, 3, 0], [1, 0, 0, 1, 1], [2, 0, 0, 0, 0]]
    # the N of the last column is neither a base nor a gap: counted in byte 5, which the DP does not score
    assert cols[:, 5].tolist() == [0, 0, 0, 0, 0, 1] and (cols[:, 6:] == 0).all()
    assert (cols[:, :6].sum(axis=1) == 3).all()
    import pytest
    with pytest.raises(ValueError):
        dp.pack_profile([b"A"] * 256)


def test_tuned_cpu_scorer_equals_the_oracle(oracle_build):
    """oracle/dp_tuned.c (bench.py's cpu_baseline.tuned: skewed two-row recurrence, 16 pairs of one shape per SIMD vector) gives
    the oracle's scores: groups of equal shape (the vector path), ragged pairs, empty profiles, a tail shorter than a vector."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    import pyoracle
    for inputs, rows in ((dp.synth_batch(11, np.full(37, 300), np.full(37, 280), 5, 5), (5, 5)),
                         (dp.synth_pairs(12, 20, 2, 150, vary_length=True), (2, 2))):
        p = dp.make_params(*rows)
        assert np.array_equal(pyoracle.dp_scores_tuned(inputs, p), pyoracle.dp_scores(inputs, p))
    la, lb = dp.ragged_lengths(13, 30, median=200, lo=1, hi=900)
    la[3] = 0
    lb[9] = 0
    inputs = dp.synth_batch(14, la, lb, 3, 6)
    p = dp.make_params(3, 6, open_per_pair=11, extend_per_pair=1)
    assert np.array_equal(pyoracle.dp_scores_tuned(inputs, p), pyoracle.dp_scores(inputs, p))
