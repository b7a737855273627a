"""The HIP DP against EVERY alignment of tiny pairs scored from the definition (tests/dp_bruteforce.py) -- not against the oracle:
the kernels' scores are the maximum over all alignments and their paths the optimum the stated order prefers, from checkpoints and
from stored decision bits.  Through the C ABI.  The hand-argued affine cases of tests/test_dp_bruteforce.py ride along."""
import numpy as np
import pytest

import dp_bruteforce as bf
from paramugsy_amd import dp
from test_dp_bruteforce import params_of, simple_sub

pytestmark = pytest.mark.gpu


def batch_of(pairs):
    ca = np.concatenate([a for a, _ in pairs] + [np.zeros((0, 8), np.uint8)])
    cb = np.concatenate([b for _, b in pairs] + [np.zeros((0, 8), np.uint8)])
    oa = np.concatenate([[0], np.cumsum([len(a) for a, _ in pairs])]).astype(np.int64)
    ob = np.concatenate([[0], np.cumsum([len(b) for _, b in pairs])]).astype(np.int64)
    return dp.DpInputs(ca, oa, cb, ob)


def gpu_align(pairs, params, path_mode):
    batch = dp.DpBatch(batch_of(pairs), params, options=dp.options(path_mode=path_mode))
    batch.run(traceback=True)
    scores, ops, n_ops = batch.fetch()
    paths = [[int(x) for x in p] for p in batch.paths(ops, n_ops)]
    batch.close()
    return [int(s) for s in scores], paths


@pytest.mark.parametrize("seed", range(6))
def test_kernels_against_every_alignment(seed):
    rng = np.random.default_rng(9100 + seed)
    for _ in range(8):  # a scoring scheme, then 40 tiny pairs under it (a batch has one set of parameters)
        _, _, sub, go, ge = bf.random_case(rng)
        pairs = []
        for _ in range(40):
            a, b, _, _, _ = bf.random_case(rng, max_len=5)
            pairs.append((a, b))
        want = [bf.best_by_enumeration(a, b, sub, go, ge)[:2] for a, b in pairs]
        for mode in (1, 2):  # stored bits; checkpoints + recomputed blocks
            scores, paths = gpu_align(pairs, params_of(sub, go, ge), mode)
            for k, (best, ops) in enumerate(want):
                assert scores[k] == best, (mode, k, sub, go, ge)
                assert paths[k] == ops, (mode, k, sub, go, ge)


def test_hand_argued_affine_cases_on_the_gpu():
    """The four cases whose optimum tests/test_dp_bruteforce.py derives by hand in its docstrings."""
    P = dp.pack_profile
    cases = [((P([b"ACGT"]), P([b"AT"])), simple_sub(2, -3), 4, 1, (-1, [0, 2, 2, 0])),
             ((P([b"A"]), P([b"C"])), simple_sub(1, -10), 4, 1, (-8, [2, 1])),
             ((P([b"A"]), P([b"C"])), simple_sub(1, -8), 4, 1, (-8, [0])),
             ((P([b"AAAA"]), P([b"A"])), simple_sub(1, -1), 2, 0, (-1, [2, 2, 2, 0])),
             ((P([b"AC", b"AC"]), P([b"A", b"A"])), simple_sub(1, -1), 3, 1, (1, [0, 2])),
             ((P([b"AC", b"AG"]), P([b"A", b"C"])), simple_sub(1, -1), 3, 1, (-3, [0, 2])),
             ((np.zeros((0, 8), np.uint8), P([b"AAA"])), simple_sub(5, -5), 1, 3, (-7, [1, 1, 1]))]
    for pair, sub, go, ge, (score, ops) in cases:
        for mode in (1, 2):
            s, p = gpu_align([pair], params_of(sub, go, ge), mode)
            assert (s[0], p[0]) == (score, ops), (sub, go, ge, mode)
