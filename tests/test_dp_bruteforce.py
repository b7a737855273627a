"""oracle/dp_oracle.c against EVERY alignment of tiny pairs, scored from the definition (tests/dp_bruteforce.py has the definition,
the labelled enumeration, the order of preference among optima and the two corners of the specification it brought out), plus
affine cases (open != extend) whose optimum is argued by hand below.  CPU only; the same pairs go through the HIP path in
tests/test_dp_bruteforce_gpu.py.  "Parity unpinned" stays (no reference DP exists: lib/maf/alignment.ml:8-10)."""
import numpy as np
import pytest

import dp_bruteforce as bf
from paramugsy_amd import dp


def params_of(sub, go, ge):
    p = dp.PmDpParams()
    for k in range(25):
        p.sub[k] = int(sub[k])
    p.gap_open, p.gap_extend = int(go), int(ge)
    return p


def inputs_of(a, b):
    return dp.DpInputs(np.ascontiguousarray(a), np.array([0, len(a)], np.int64), np.ascontiguousarray(b), np.array([0, len(b)], np.int64))


def oracle_align(a, b, sub, go, ge):
    import pyoracle
    scores, paths = pyoracle.dp_align(inputs_of(a, b), params_of(sub, go, ge))
    return int(scores[0]), [int(x) for x in paths[0]]


@pytest.mark.parametrize("seed", range(8))
def test_every_alignment_of_tiny_pairs(seed, oracle_build):
    """250 random tiny pairs per seed (lengths 0..5, 1..3 rows a side, random 5 x 5 matrices, go != ge incl. zero penalties and
    go < ge): the oracle's score is the maximum over ALL alignments scored from the definition, and its path is the one optimum the
    stated order prefers.  Most cases have several optima (the generator is tie-heavy), so the order is exercised, not assumed."""
    rng = np.random.default_rng(9000 + seed)
    several = 0
    for _ in range(250):
        a, b, sub, go, ge = bf.random_case(rng, max_len=5, tie_heavy=rng.random() < 0.8)
        best, ops, n_opt = bf.best_by_enumeration(a, b, sub, go, ge)
        score, path = oracle_align(a, b, sub, go, ge)
        assert score == best, (a[:, :5].tolist(), b[:, :5].tolist(), sub, go, ge)
        assert path == ops, (a[:, :5].tolist(), b[:, :5].tolist(), sub, go, ge, n_opt)
        several += n_opt > 1
    assert several > 30  # the tie order was put to the test (a fifth to a quarter of the cases have several optima)


def test_six_by_six(oracle_build):
    """A few pairs at the size VERDICT r4 names (La, Lb <= 6): 8 989 unlabelled alignments of a 6 x 6 pair, all written out."""
    rng = np.random.default_rng(66)
    for _ in range(6):
        a, b, sub, go, ge = bf.random_case(rng, max_len=6, tie_heavy=True)
        while len(a) < 5 or len(b) < 5:
            a, b, sub, go, ge = bf.random_case(rng, max_len=6, tie_heavy=True)
        best, ops, _ = bf.best_by_enumeration(a, b, sub, go, ge)
        assert oracle_align(a, b, sub, go, ge) == (best, ops)


def test_the_enumeration_counts_what_it_should():
    """Delannoy numbers: the unlabelled alignments of an m x n pair (1, 3, 13, 63, 321, 1683, 8989 on the diagonal)."""
    assert [len(bf.plain_alignments(n, n)) for n in range(6)] == [1, 3, 13, 63, 321, 1683]
    assert len(bf.plain_alignments(2, 5)) == 61 and len(bf.plain_alignments(0, 4)) == 1
    # labelled strings project onto every unlabelled one
    for la, lb in ((2, 3), (3, 3), (0, 2), (4, 1)):
        lab = {tuple(bf.OP_OF[x] for x in s) for s in bf.labelled_alignments(la, lb)}
        assert lab == set(bf.plain_alignments(la, lb))


SIMPLE = [0] * 25


def simple_sub(match, mismatch, gap=0):
    sub = [0] * 25
    for x in range(5):
        for y in range(5):
            sub[x * 5 + y] = 0 if (x == 4 and y == 4) else gap if (x == 4 or y == 4) else (match if x == y else mismatch)
    return sub


def test_affine_by_hand_one_long_gap_beats_two_short_ones(oracle_build):
    """A = ACGT, B = AT (one row each), match +2, mismatch -3, open 4, extend 1.  B is two columns shorter: every alignment holds two
    D's (or more gap columns, which only cost).  With exactly two D's and two M's:
        M D D M   A/A, C-, G-, T/T   2 + 2 - (4 + 1)      = -1     one run of two
        M D M D   A/A, C-, G/T, T-   2 - 4 - 3 - 4        = -9     two runs, and a mismatch
        D M D M   A-, C/A, G-, T/T   -4 - 3 - 4 + 2       = -9
        D D M M   A-, C-, G/A, T/T   -(4 + 1) - 3 + 2     = -6
        M M D D   A/A, C/T, G-, T-   2 - 3 - (4 + 1)      = -6
        D M M D   A-, C/A, G/T, T-   -4 - 3 - 3 - 4       = -14
    Alignments with an I need a third D: at best 2 + 2 minus three gap columns in at least two runs <= 4 - (4 + 4 + 1) = -5.
    So the optimum is M D D M = -1, and it is unique.  Under LINEAR gaps of 4 the same alignment costs 2 + 2 - 8 = -4: the test also
    holds that the extension price is what made the difference."""
    a, b = dp.pack_profile([b"ACGT"]), dp.pack_profile([b"AT"])
    assert oracle_align(a, b, simple_sub(2, -3), 4, 1) == (-1, [0, 2, 2, 0])
    assert bf.best_by_enumeration(a, b, simple_sub(2, -3), 4, 1) == (-1, [0, 2, 2, 0], 1)
    assert oracle_align(a, b, simple_sub(2, -3), 4, 4)[0] == -4


def test_affine_by_hand_an_insertion_directly_followed_by_a_deletion(oracle_build):
    """A = A, B = C, mismatch -10, open 4, extend 1.  Three alignments exist: M (-10), I D and D I (two runs of one: -4 - 4 = -8 each;
    an I run directly followed by a D run pays two openings -- there is no cheaper "mixed" run).  -8 > -10, two optima; read from the
    end, D I ends in an insertion and I D in a deletion: the order M < I < D reports D I."""
    a, b = dp.pack_profile([b"A"]), dp.pack_profile([b"C"])
    assert oracle_align(a, b, simple_sub(1, -10), 4, 1) == (-8, [2, 1])
    assert bf.best_by_enumeration(a, b, simple_sub(1, -10), 4, 1) == (-8, [2, 1], 2)
    # with a mismatch of -8 the three tie, and the match wins the tie
    assert oracle_align(a, b, simple_sub(1, -8), 4, 1) == (-8, [0])


def test_affine_by_hand_free_extension_and_where_the_run_goes(oracle_build):
    """A = AAAA, B = A, match +1, open 2, extend 0.  One M and three D's in one run: M D D D = D D D M = 1 - 2 = -1; with the M in the
    middle the D's are two runs: D M D D = D D M D = 1 - 2 - 2 = -3; no M at all: I D D D D etc. <= -2 - 2 = -4.  Two optima; read from
    the end, D D D M ends in a match: it is the one reported."""
    a, b = dp.pack_profile([b"AAAA"]), dp.pack_profile([b"A"])
    assert oracle_align(a, b, simple_sub(1, -1), 2, 0) == (-1, [2, 2, 2, 0])
    assert bf.best_by_enumeration(a, b, simple_sub(1, -1), 2, 0) == (-1, [2, 2, 2, 0], 2)


def test_affine_by_hand_profiles_of_two_rows(oracle_build):
    """Sum of pairs over rows.  A = rows AC / AC (columns {A:2}, {C:2}), B = rows A / A (column {A:2}); match +1, mismatch -1, open 3,
    extend 1.  s({A:2}, {A:2}) = 2 * 2 * (+1) = 4, s({C:2}, {A:2}) = 2 * 2 * (-1) = -4.  A is one column longer:
        M D  = 4 - 3 = 1        D M  = -3 - 4 = -7        I D D = D I D = D D I = -3 - (3 + 1) = -7 (two runs)
    so M D with score 1.  With a mixed column, A = rows AC / AG, the second column is {C:1, G:1}: s = (1 + 1) * 2 * (-1) = -4 still;
    and B = rows A / C against A's first column {A:2}: 2 * (1 * 1 + 1 * (-1)) = 0, so M D scores 0 - 3 = -3 and D M scores -3 + s({C:1, G:1},
    {A:1, C:1}) = -3 + (1 * (-1) + 1 * 1 + 1 * (-1) + 1 * (-1)) = -5: M D still."""
    sub = simple_sub(1, -1)
    a, b = dp.pack_profile([b"AC", b"AC"]), dp.pack_profile([b"A", b"A"])
    assert oracle_align(a, b, sub, 3, 1) == (1, [0, 2])
    a2, b2 = dp.pack_profile([b"AC", b"AG"]), dp.pack_profile([b"A", b"C"])
    assert oracle_align(a2, b2, sub, 3, 1) == (-3, [0, 2])
    assert bf.best_by_enumeration(a2, b2, sub, 3, 1)[:2] == (-3, [0, 2])


def test_the_boundary_is_one_run_also_when_an_extension_is_dearer_than_an_opening(oracle_build):
    """The corner tests/dp_bruteforce.py describes: go = 1 < ge = 3.  A is empty, B = AAA: the only alignment is I I I, and the
    specification prices the boundary as ONE run, 1 + 2 * 3 = 7 (not three openings, 3).  Off the boundary re-opening is allowed:
    A = A, B = AAAA with match +5: M I I I = 5 - (1 + 1 + 1) = 2 beats every alignment that starts with insertions
    (I M I I: the leading I is on the boundary, one opening; then two more openings: 5 - 1 - 1 - 1 = 2 as well -- a tie, and read from the
    end both end in I I opening ... the enumeration settles which; the oracle must agree)."""
    a0 = np.zeros((0, 8), np.uint8)
    b3 = dp.pack_profile([b"AAA"])
    assert oracle_align(a0, b3, simple_sub(5, -5), 1, 3) == (-7, [1, 1, 1])
    a1, b4 = dp.pack_profile([b"A"]), dp.pack_profile([b"AAAA"])
    best, ops, _ = bf.best_by_enumeration(a1, b4, simple_sub(5, -5), 1, 3)
    assert best == 2
    assert oracle_align(a1, b4, simple_sub(5, -5), 1, 3) == (best, ops)
